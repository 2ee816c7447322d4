// image_io.hpp -- the file formats image_warping's harness reads: 8-bit non-interlaced PNG (the reference
// loads them with LodePNG, examples/image_warping/src/main.cpp:80-81) through zlib, and the .constraints marker list
// (main.cpp:4-27: count, then x y target_x target_y per marker).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace harness {

struct Image8 {
    unsigned width = 0, height = 0, channels = 0;
    std::vector<uint8_t> px;                         // row-major, interleaved channels
    uint8_t at(unsigned x, unsigned y, unsigned c = 0) const { return px[((size_t)y * width + x) * channels + c]; }
};

namespace detail {
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace detail

inline Image8 read_png(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if (b.size() < 8 || std::memcmp(b.data(), sig, 8) != 0) throw std::runtime_error(path + ": not a PNG");
    Image8 im;
    std::vector<uint8_t> idat;
    unsigned depth = 0, ctype = 0, interlace = 0;
    for (size_t pos = 8; pos + 12 <= b.size();) {
        const uint32_t len = detail::be32(&b[pos]);
        const std::string typ(reinterpret_cast<const char*>(&b[pos + 4]), 4);
        if ((size_t)len > b.size() - pos - 12) throw std::runtime_error(path + ": PNG chunk runs past the end of the file");
        const uint8_t* d = &b[pos + 8];
        if (typ == "IHDR") { if (len < 13) throw std::runtime_error(path + ": short IHDR"); im.width = detail::be32(d); im.height = detail::be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (typ == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (typ == "IEND") break;
        pos += 12 + (size_t)len;
    }
    im.channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (depth != 8 || !im.channels || interlace) throw std::runtime_error(path + ": only 8-bit non-interlaced gray/RGB/RGBA PNGs are supported");
    // (deflate expands at most ~1032 : 1: an image the IDAT bytes cannot possibly hold is refused before it is allocated)
    if (!im.width || !im.height || (double)im.width * im.height * im.channels > 1100.0 * (double)idat.size() + 65536.0) throw std::runtime_error(path + ": PNG dimensions do not fit its data");
    const size_t stride = (size_t)im.width * im.channels;
    std::vector<uint8_t> raw((stride + 1) * im.height);
    uLongf out_len = raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), idat.size()) != Z_OK || out_len != raw.size()) throw std::runtime_error(path + ": bad zlib stream");
    im.px.resize(stride * im.height);
    const int C = im.channels;
    for (unsigned y = 0; y < im.height; ++y) {
        const uint8_t ft = raw[y * (stride + 1)];
        const uint8_t* line = &raw[y * (stride + 1) + 1];
        uint8_t* cur = &im.px[y * stride];
        const uint8_t* prev = y ? &im.px[(y - 1) * stride] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)C ? cur[i - C] : 0, up = prev ? prev[i] : 0, ul = (prev && i >= (size_t)C) ? prev[i - C] : 0;
            int pred = 0;
            switch (ft) { case 1: pred = a; break; case 2: pred = up; break; case 3: pred = (a + up) >> 1; break; case 4: pred = detail::paeth(a, up, ul); break; default: break; }
            cur[i] = (uint8_t)(line[i] + pred);
        }
    }
    return im;
}

inline void write_png(const std::string& path, const Image8& im)
{
    const size_t stride = (size_t)im.width * im.channels;
    std::vector<uint8_t> raw;
    raw.reserve((stride + 1) * im.height);
    for (unsigned y = 0; y < im.height; ++y) { raw.push_back(0); raw.insert(raw.end(), &im.px[y * stride], &im.px[y * stride] + stride); }
    uLongf zl = compressBound(raw.size());
    std::vector<uint8_t> z(zl);
    if (compress2(z.data(), &zl, raw.data(), raw.size(), 6) != Z_OK) throw std::runtime_error("zlib compress failed");
    z.resize(zl);
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    auto chunk = [&](const char* typ, const std::vector<uint8_t>& data) {
        detail::put32(out, (uint32_t)data.size());
        const size_t start = out.size();
        out.insert(out.end(), typ, typ + 4); out.insert(out.end(), data.begin(), data.end());
        detail::put32(out, (uint32_t)crc32(0L, &out[start], (uInt)(out.size() - start)));
    };
    std::vector<uint8_t> hdr;
    detail::put32(hdr, im.width); detail::put32(hdr, im.height);
    hdr.push_back(8); hdr.push_back(im.channels == 1 ? 0 : im.channels == 3 ? 2 : im.channels == 2 ? 4 : 6); hdr.push_back(0); hdr.push_back(0); hdr.push_back(0);
    chunk("IHDR", hdr); chunk("IDAT", z); chunk("IEND", {});
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)out.size());
}

// count, then 4 integers per marker: x y target_x target_y
inline std::vector<std::vector<int>> read_constraints(const std::string& path)
{
    std::ifstream in(path);
    if (!in.good()) throw std::runtime_error("could not open marker file " + path);
    unsigned n = 0;
    in >> n;
    {   std::ifstream sz(path, std::ios::binary | std::ios::ate);
        if (!in || (size_t)n > (size_t)sz.tellg() / 8) throw std::runtime_error(path + ": bad marker count"); }      // ("0 0 0 0\n": 8 bytes a marker)
    std::vector<std::vector<int>> c(n, std::vector<int>(4, 0));
    for (auto& m : c) for (int& v : m) in >> v;
    if (!in) throw std::runtime_error(path + ": truncated marker list");
    return c;
}

}  // namespace harness
