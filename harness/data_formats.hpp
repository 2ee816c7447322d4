// data_formats.hpp -- the remaining input formats of the reference's example applications, host C++:
//   .imagedump   int32 width, height, channels, datatype (0 float32, 1 uint8) + raw rows   examples/shape_from_shading/src/SimpleBuffer.cpp:12-52
//   *.SFSSolverParameters   160-byte struct dump                                            examples/shape_from_shading/src/TerraSolverParameters.h:7-45
//   .off / .ply  triangle meshes (ascii OFF; ascii / binary_little_endian PLY with x y z + a face index list)  -- read through OpenMesh there
//   .mrk         landmarks: count, then "x y z radius vertex_index" per line               examples/arap_mesh_deformation/src/LandMarkSet.h
// (python twins: thallo_amd/formats.py; the files shipped with the reference were read with both.)
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace harness {

// Every count a file states is checked against the bytes the file actually has before anything is sized by it (tools/formats_fuzz.cpp mutates the fixtures
// under AddressSanitizer: a count of 2^60 vertices must be a message, not an allocation)
inline size_t file_bytes(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    return f.good() ? (size_t)f.tellg() : 0;
}

struct ImageDump {
    int width = 0, height = 0, channels = 0, datatype = 0;           // datatype 0: float32, 1: uint8
    std::vector<float> f; std::vector<uint8_t> u;
};

inline ImageDump read_imagedump(const std::string& path, bool clamp_infinity = true)
{
    std::ifstream in(path, std::ios::binary);
    if (!in.good()) throw std::runtime_error("cannot open " + path);
    ImageDump d; int32_t h[4];
    in.read(reinterpret_cast<char*>(h), 16);
    d.width = h[0]; d.height = h[1]; d.channels = h[2]; d.datatype = h[3];
    if (!in || d.width <= 0 || d.height <= 0 || d.channels <= 0 || d.channels > 64 || (d.datatype != 0 && d.datatype != 1)) throw std::runtime_error(path + ": bad imagedump header");
    const size_t n = (size_t)d.width * (size_t)d.height * (size_t)d.channels;
    if (n > file_bytes(path) || n * (d.datatype == 0 ? 4 : 1) + 16 > file_bytes(path)) throw std::runtime_error(path + ": truncated imagedump");
    if (d.datatype == 0) {
        d.f.resize(n); in.read(reinterpret_cast<char*>(d.f.data()), (std::streamsize)(n * 4));
        if (clamp_infinity)                                           // SimpleBuffer.cpp:29-41: +inf -> FLT_MAX, -inf -> -10000 (first w*h floats)
            for (size_t i = 0; i < (size_t)d.width * d.height; ++i)
                if (std::isinf(d.f[i])) d.f[i] = d.f[i] > 0 ? std::numeric_limits<float>::max() : -10000.0f;
    } else { d.u.resize(n); in.read(reinterpret_cast<char*>(d.u.data()), (std::streamsize)n); }
    if (!in) throw std::runtime_error(path + ": truncated imagedump");
    return d;
}

inline void write_imagedump(const std::string& path, int w, int h, const std::vector<float>& data)
{
    std::ofstream out(path, std::ios::binary);
    const int32_t hd[4] = { w, h, 1, 0 };
    out.write(reinterpret_cast<const char*>(hd), 16);
    out.write(reinterpret_cast<const char*>(data.data()), (std::streamsize)(data.size() * 4));
}

struct SfsParameters {          // the 160 bytes of TerraSolverParameters
    float weightFitting, weightRegularizer, weightPrior, weightShading, weightShadingStart, weightShadingIncrement, weightBoundary;
    float fx, fy, ux, uy;
    float deltaTransform[16];
    float lightingCoefficients[9];
    unsigned int unused[3];
    unsigned int pad;
};
static_assert(sizeof(SfsParameters) == 160, "SFS parameter struct layout");

inline SfsParameters read_sfs_parameters(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    SfsParameters p; std::memset(&p, 0, sizeof(p));
    in.read(reinterpret_cast<char*>(&p), sizeof(p));
    if (in.gcount() < 156) throw std::runtime_error(path + ": expected a 160-byte parameter file");
    return p;
}

struct Mesh { std::vector<std::array<float, 3>> v; std::vector<std::vector<int>> f; };

inline Mesh read_off(const std::string& path)
{
    std::ifstream in(path);
    std::string magic; size_t nv = 0, nf = 0, ne = 0;
    in >> magic >> nv >> nf >> ne;
    if (!in || magic != "OFF") throw std::runtime_error(path + ": not an OFF file");
    const size_t bytes = file_bytes(path);
    if (nv > bytes / 6 || nf > bytes / 2) throw std::runtime_error(path + ": the OFF header counts more vertices / faces than the file can hold");       // ("0 0 0\n": 6 bytes a vertex)
    Mesh m; m.v.resize(nv); m.f.resize(nf);
    for (auto& p : m.v) in >> p[0] >> p[1] >> p[2];
    for (auto& f : m.f) {
        long k = 0; in >> k;
        if (!in || k < 0 || (size_t)k > bytes / 2) throw std::runtime_error(path + ": bad face in OFF file");
        f.resize((size_t)k); for (int& i : f) in >> i;
    }
    if (!in) throw std::runtime_error(path + ": truncated OFF file");
    for (auto& f : m.f) for (int i : f) if (i < 0 || (size_t)i >= nv) throw std::runtime_error(path + ": face index out of range");
    return m;
}

inline Mesh read_ply(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in.good()) throw std::runtime_error("cannot open " + path);
    struct Prop { std::string type, ctype, itype, name; bool list; };
    struct Elem { std::string name; size_t n; std::vector<Prop> props; };
    std::vector<Elem> elems; std::string fmt, line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line); std::string t; ls >> t;
        if (t == "format") ls >> fmt;
        else if (t == "element") { Elem e; e.n = 0; ls >> e.name >> e.n; if (!ls) throw std::runtime_error(path + ": bad PLY element line"); elems.push_back(e); }
        else if (t == "property") {
            Prop p; ls >> p.type; p.list = p.type == "list"; if (p.list) ls >> p.ctype >> p.itype; ls >> p.name;
            if (elems.empty()) throw std::runtime_error(path + ": PLY property in front of the first element");
            elems.back().props.push_back(p);
        }
        else if (t == "end_header") break;
    }
    auto size_of = [](const std::string& t) -> int {
        if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
        if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
        if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
        if (t == "double" || t == "float64") return 8;
        throw std::runtime_error("unknown PLY type " + t);
    };
    Mesh m;
    const bool ascii = fmt == "ascii";
    if (!ascii && fmt != "binary_little_endian") throw std::runtime_error(path + ": unsupported PLY format " + fmt);
    auto read_num = [&](const std::string& t) -> double {
        if (ascii) { double v; in >> v; return v; }
        char b[8]; const int n = size_of(t); in.read(b, n);
        if (t == "float" || t == "float32") { float v; std::memcpy(&v, b, 4); return v; }
        if (t == "double" || t == "float64") { double v; std::memcpy(&v, b, 8); return v; }
        if (n == 1) return t[0] == 'u' ? (double)(uint8_t)b[0] : (double)(int8_t)b[0];
        if (n == 2) { int16_t v; std::memcpy(&v, b, 2); return t[0] == 'u' ? (double)(uint16_t)v : (double)v; }
        int32_t v; std::memcpy(&v, b, 4); return t[0] == 'u' ? (double)(uint32_t)v : (double)v;
    };
    const size_t bytes = file_bytes(path);
    for (auto& e : elems) {
        if (e.n > bytes) throw std::runtime_error(path + ": the PLY header counts more " + e.name + " elements than the file has bytes");
        if (e.props.empty() && e.n) throw std::runtime_error(path + ": PLY element " + e.name + " without properties");
        for (size_t i = 0; i < e.n; ++i) {
            std::array<float, 3> p = { 0, 0, 0 }; std::vector<int> face;
            if (!in) throw std::runtime_error(path + ": truncated PLY file");
            for (auto& pr : e.props) {
                if (pr.list) {
                    const double kd = read_num(pr.ctype);
                    if (!in || !(kd >= 0) || kd > (double)bytes) throw std::runtime_error(path + ": bad PLY list length");
                    std::vector<int> l((size_t)kd); for (int& x : l) x = (int)read_num(pr.itype); face = l;
                }
                else { const double v = read_num(pr.type); if (pr.name == "x") p[0] = (float)v; else if (pr.name == "y") p[1] = (float)v; else if (pr.name == "z") p[2] = (float)v; }
            }
            if (e.name == "vertex") m.v.push_back(p); else if (e.name == "face") m.f.push_back(face);
        }
    }
    if (!in && !in.eof()) throw std::runtime_error(path + ": truncated PLY file");
    for (auto& f : m.f) for (int i : f) if (i < 0 || (size_t)i >= m.v.size()) throw std::runtime_error(path + ": face index out of range");
    return m;
}

inline Mesh read_mesh(const std::string& path)
{
    const std::string ext = path.size() >= 4 ? path.substr(path.size() - 4) : "";
    if (ext == ".off" || ext == ".OFF") return read_off(path);
    return read_ply(path);
}

inline void write_ply_ascii(const std::string& path, const Mesh& m)
{
    std::ofstream out(path);
    out << "ply\nformat ascii 1.0\nelement vertex " << m.v.size() << "\nproperty float x\nproperty float y\nproperty float z\nelement face " << m.f.size()
        << "\nproperty list uchar int vertex_indices\nend_header\n";
    for (auto& p : m.v) out << p[0] << " " << p[1] << " " << p[2] << "\n";
    for (auto& f : m.f) { out << f.size(); for (int i : f) out << " " << i; out << "\n"; }
}

struct Landmarks { std::vector<int> index; std::vector<std::array<float, 3>> target; };

inline Landmarks read_mrk(const std::string& path)
{
    std::ifstream in(path);
    if (!in.good()) throw std::runtime_error("could not open marker file " + path);
    size_t n = 0; in >> n;
    if (!in || n > file_bytes(path) / 8) throw std::runtime_error(path + ": bad landmark count");        // ("0 0 0 0 0\n": 10 bytes a landmark)
    Landmarks l; l.index.resize(n); l.target.resize(n);
    for (size_t i = 0; i < n; ++i) { float radius; in >> l.target[i][0] >> l.target[i][1] >> l.target[i][2] >> radius >> l.index[i]; }
    if (!in) throw std::runtime_error(path + ": truncated landmark list");
    return l;
}

// both directions of every undirected mesh edge, grouped by the first vertex in vertex order: the one-ring layout of
// examples/arap_mesh_deformation/src/CombinedSolver.h:120-150
inline void directed_edges(const Mesh& m, std::vector<int>& v0, std::vector<int>& v1)
{
    std::set<std::pair<int, int>> und;
    for (auto& f : m.f) for (int i : f) if (i < 0 || (size_t)i >= m.v.size()) throw std::runtime_error("mesh face index out of range");
    for (auto& f : m.f) for (size_t i = 0; i < f.size(); ++i) { const int a = f[i], b = f[(i + 1) % f.size()]; if (a != b) und.insert({ std::min(a, b), std::max(a, b) }); }
    std::vector<std::vector<int>> nbr(m.v.size());
    for (auto& e : und) { nbr[e.first].push_back(e.second); nbr[e.second].push_back(e.first); }
    v0.clear(); v1.clear();
    for (size_t v = 0; v < nbr.size(); ++v) { std::sort(nbr[v].begin(), nbr[v].end()); for (int w : nbr[v]) { v0.push_back((int)v); v1.push_back(w); } }
}

}  // namespace harness
