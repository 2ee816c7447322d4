// frontend.cpp -- recognise a bundled energy from its .t problem specification.
//
// The reference executes the .t in a Lua/Terra sandbox (API/src/thallo.t:1359-1434, lib.t:584-593),
// differentiates it symbolically and JIT-compiles kernels.  This backend has no Lua VM: it reads the
// declarative surface of the file -- Dims(), the Inputs{} table (API/src/thallo.t:1580-2112 for the
// constructor names), UsePreconditioner(), the Residuals{} keys and numeric literals -- and selects
// the hand-written gfx950 plugin with the same signature.  A file whose signature matches but whose
// (comment/whitespace-stripped) text is not one of the known bundled files is accepted with a
// warning; a file with no matching signature is rejected (Plan returns NULL, as the reference does
// when compilation fails: thallo.t:1431-1432).
#include "plugin.hpp"
#include "dsl.hpp"
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <set>
#include <sstream>

namespace thallo {

namespace {

std::string strip_comments(const std::string& s)
{
    std::string o; o.reserve(s.size());
    size_t i = 0;
    if (s.size() >= 3 && (unsigned char)s[0] == 0xEF && (unsigned char)s[1] == 0xBB && (unsigned char)s[2] == 0xBF) i = 3;   // BOM
    while (i < s.size()) {
        if (s[i] == '"' || s[i] == '\'') {                       // string literal
            const char q = s[i]; o += s[i++];
            while (i < s.size() && s[i] != q) { if (s[i] == '\\' && i + 1 < s.size()) o += s[i++]; o += s[i++]; }
            if (i < s.size()) o += s[i++];
        } else if (s[i] == '-' && i + 1 < s.size() && s[i + 1] == '-') {
            size_t j = i + 2;
            if (j < s.size() && s[j] == '[') {                   // long comment --[[ ]] or --[=[ ]=]
                size_t k = j + 1; int eq = 0;
                while (k < s.size() && s[k] == '=') { ++eq; ++k; }
                if (k < s.size() && s[k] == '[') {
                    std::string close = "]" + std::string(eq, '=') + "]";
                    size_t e = s.find(close, k + 1);
                    i = (e == std::string::npos) ? s.size() : e + close.size();
                    o += ' ';
                    continue;
                }
            }
            while (i < s.size() && s[i] != '\n') ++i;              // line comment
        } else o += s[i++];
    }
    return o;
}

std::string squeeze(const std::string& s)
{
    std::string o; for (char c : s) if (!isspace((unsigned char)c)) o += c; return o;
}

unsigned long long fnv1a(const std::string& s)
{
    unsigned long long h = 1469598103934665603ULL;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ULL; }
    return h;
}

struct Tok { enum K { ID, NUM, STR, PUNCT, END } k; std::string s; };

std::vector<Tok> lex(const std::string& s)
{
    std::vector<Tok> t; size_t i = 0;
    while (i < s.size()) {
        const unsigned char c = s[i];
        if (isspace(c)) { ++i; continue; }
        if (isalpha(c) || c == '_') { size_t j = i; while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '_')) ++j; t.push_back({ Tok::ID, s.substr(i, j - i) }); i = j; }
        else if (isdigit(c) || (c == '.' && i + 1 < s.size() && isdigit((unsigned char)s[i + 1]))) {
            size_t j = i; while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '.' || ((s[j] == '-' || s[j] == '+') && (s[j - 1] == 'e' || s[j - 1] == 'E')))) ++j;
            t.push_back({ Tok::NUM, s.substr(i, j - i) }); i = j;
        } else if (c == '"' || c == '\'') { size_t j = i + 1; while (j < s.size() && s[j] != (char)c) ++j; t.push_back({ Tok::STR, s.substr(i + 1, j - i - 1) }); i = j + 1; }
        else { t.push_back({ Tok::PUNCT, std::string(1, (char)c) }); ++i; }
    }
    t.push_back({ Tok::END, "" });
    return t;
}

struct InputDecl { std::string name, kind, type; int index = -1; int channels = 0; };

int type_channels(const std::string& ty)
{
    if (ty == "uint8" || ty == "uchar" || ty == "bool") return 1;
    size_t i = ty.size(); while (i > 0 && isdigit((unsigned char)ty[i - 1])) --i;
    const std::string base = ty.substr(0, i), num = ty.substr(i);
    if (base != "float" && base != "thallo_float" && base != "double" && base != "uint8" && base != "uchar" && base != "int") return 0;
    return num.empty() ? 1 : atoi(num.c_str());
}

std::string sig_of(const std::vector<InputDecl>& in)
{
    std::vector<std::string> by(in.size());
    std::ostringstream o;
    std::vector<const InputDecl*> ord(in.size(), nullptr);
    for (auto& d : in) if (d.index >= 0 && d.index < (int)in.size()) ord[d.index] = &d;
    for (auto* d : ord) { if (!d) { o << "?;"; continue; } o << d->kind[0] << d->channels << ";"; }
    return o.str();
}

struct Known { const char* energy; int n_dims; const char* sig; const char* keys; bool precond; };
const Known KNOWN[] = {
    // signature: per input index: <Kind initial><channels>;   U=Unknown A=Array S=Sparse P=Param
    { "laplacian_image", 2, "U1;A1;",                         "fit,reg",                        false },
    { "laplacian_graph", 2, "U1;A1;S0;S0;",                   "fit,reg",                        false },
    { "image_warping",   2, "U2;U1;A2;A2;A1;P1;P1;",          "fit,reg_nx,reg_ny,reg_px,reg_py", true },
    { "arap_mesh",       2, "P1;P1;U3;U3;A3;A3;S0;S0;",       "fit,reg",                        true },
    { "shape_from_shading", 2, "P1;P1;P1;P1;P1;P1;P1;P1;P1;P1;P1;P1;P1;P1;P1;P1;U1;A1;A1;A1;A1;", "fit,reg,shading_h,shading_v", false },
    { "bundle_adjustment", 3, "U9;U3;A2;S0;S0;",               "snavely_reprojection_error",     true },
};

#include "known_energy_hashes.inc"

}  // namespace

bool parse_problem_file(const char* filename, ProblemSpec& out)
{
    out = ProblemSpec(); out.file = filename ? filename : "";
    std::ifstream f(out.file, std::ios::binary);
    if (!f) { out.diagnostic = "cannot open problem specification '" + out.file + "'"; return false; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string text = strip_comments(ss.str());
    const std::string packed = squeeze(text);
    const unsigned long long h = fnv1a(packed);
    out.body_hash = h;
    const std::vector<Tok> t = lex(text);

    std::vector<InputDecl> inputs; std::set<std::string> keys; bool precond = false;
    for (size_t i = 0; i + 1 < t.size(); ++i) {
        if (t[i].k == Tok::ID && t[i].s == "Dims" && t[i + 1].s == "(") {
            size_t j = i + 2; int n = 0; while (t[j].k != Tok::END && t[j].s != ")") { if (t[j].k == Tok::STR) ++n; ++j; }
            out.n_dims = n;
        }
        if (t[i].k == Tok::ID && t[i].s == "UsePreconditioner" && t[i + 1].s == "(" && t[i + 2].s == "true") precond = true;
        if (t[i].k == Tok::ID && t[i].s == "Inputs" && t[i + 1].s == "{") {
            size_t j = i + 2;
            while (t[j].k != Tok::END && t[j].s != "}") {
                if (t[j].k == Tok::ID && t[j + 1].s == "=" && t[j + 2].k == Tok::ID && t[j + 3].s == "(") {
                    InputDecl d; d.name = t[j].s; d.kind = t[j + 2].s;
                    size_t k = j + 4; int depth = 1; int brace = 0; std::string last_num;
                    while (t[k].k != Tok::END && depth > 0) {
                        if (t[k].s == "(") ++depth; else if (t[k].s == ")") --depth;
                        else if (t[k].s == "{") ++brace; else if (t[k].s == "}") --brace;
                        else if (t[k].k == Tok::ID && brace == 0 && d.type.empty() && depth == 1) d.type = t[k].s;
                        else if (t[k].k == Tok::NUM && brace == 0 && depth == 1) last_num = t[k].s;
                        ++k;
                    }
                    if (!last_num.empty()) d.index = atoi(last_num.c_str());
                    d.channels = (d.kind == "Sparse") ? 0 : type_channels(d.type);
                    if (d.kind == "Param") d.channels = 1;
                    inputs.push_back(d);
                    j = k;
                } else ++j;
            }
        }
        if (t[i].k == Tok::ID && t[i].s == "Residuals" && t[i + 1].s == "{") {
            size_t j = i + 2; int depth = 1, par = 0;
            while (t[j].k != Tok::END && depth > 0) {
                if (t[j].s == "{") ++depth; else if (t[j].s == "}") --depth;
                else if (t[j].s == "(") ++par; else if (t[j].s == ")") --par;
                else if (depth == 1 && par == 0 && t[j].k == Tok::ID && t[j + 1].s == "=" && t[j + 2].s != "=") keys.insert(t[j].s);
                ++j;
            }
        }
        // numeric literals bound to a name at statement level:  [local] name = 0.5
        if (t[i].k == Tok::ID && t[i + 1].s == "=" && t[i + 2].k == Tok::NUM && t[i + 3].s != "*" && t[i + 3].s != "+" &&
            t[i + 3].s != "-" && t[i + 3].s != "/" && (i == 0 || (t[i - 1].s != "," && t[i - 1].s != "{" && t[i - 1].s != "(")))
            out.constants[t[i].s] = atof(t[i + 2].s.c_str());
    }
    {   // schedule lines:  <residuals>.<name>.J:set_materialize(true)  /  .JtJ:set_materialize(true)   (thallo.t:5661-5690)
        std::set<std::string> matJ, matJtJ;
        for (size_t i = 0; i + 8 < t.size(); ++i) {
            if (t[i].k == Tok::ID && t[i + 1].s == "." && t[i + 2].k == Tok::ID && t[i + 3].s == "." && t[i + 4].k == Tok::ID &&
                t[i + 5].s == ":" && t[i + 6].s == "set_materialize" && t[i + 7].s == "(" && t[i + 8].s == "true") {
                if (t[i + 4].s == "J") matJ.insert(t[i + 2].s);
                if (t[i + 4].s == "JtJ") matJtJ.insert(t[i + 2].s);
            }
        }
        // honoured when it covers every residual group (a mixed schedule runs matrix-free: same values, different evaluation order)
        if (!keys.empty() && matJ.size() == keys.size()) out.constants["materialize_J"] = 1.0;
        if (!keys.empty() && matJtJ.size() == keys.size()) out.constants["materialize_JtJ"] = 1.0;
        // a schedule the hand-written plugins do not implement (only some residuals materialized, e.g. tests/minimal_graph/laplacian.t:19-20,
        // or a materialized Jp): the Plan hands the file to the front-end, whose generated plugin schedules per residual (dsl_plugin.cpp)
        bool any_jp = false;
        for (size_t i = 0; i + 4 < t.size(); ++i) if (t[i].s == "Jp" && t[i + 1].s == ":" && t[i + 2].s == "set_materialize" && t[i + 4].s == "true") any_jp = true;
        if (any_jp || (!matJ.empty() && matJ.size() != keys.size()) || (!matJtJ.empty() && matJtJ.size() != keys.size())) out.constants["schedule_per_residual"] = 1.0;
    }
    std::string keystr; for (auto& k : keys) { if (!keystr.empty()) keystr += ","; keystr += k; }
    const std::string sig = sig_of(inputs);
    for (const Known& k : KNOWN) {
        if (sig == k.sig && keystr == k.keys && precond == k.precond && out.n_dims == k.n_dims) { out.energy = k.energy; break; }
    }
    if (out.energy.empty()) {
        out.diagnostic = "'" + out.file + "': no bundled gfx950 plugin matches this problem specification (inputs " + sig +
                         " residuals {" + keystr + "}); this backend runs precompiled energies only";
        return false;
    }
    if (out.energy == "laplacian_image") {
        // x-neighbour guard as written in the file: InBounds(x+1,y+1) (shipped) or InBounds(x+1,y)
        out.constants["xguard"] = packed.find("InBounds(x+1,y+1)") != std::string::npos ? 0.0 : 1.0;
    }
    for (unsigned long long kh : KNOWN_BODY_HASHES) if (kh == h) out.verified_body = true;
    if (!out.verified_body) {
        char buf[64]; snprintf(buf, sizeof(buf), "%016llx", h);
        out.diagnostic = "'" + out.file + "' matched plugin '" + out.energy + "' by signature only (body hash " + buf +
                         " is not a known bundled file)";
    }
    return true;
}

}  // namespace thallo

// 0 = matrix-free, 1 = `[Jt][[J]p]`, 2 = `[[Jt][J]]p` as requested by the file's set_materialize lines; -1 = no plugin for the file
extern "C" int ThalloX_ProblemFileSchedule(const char* filename)
{
    thallo::ProblemSpec spec;
    if (!thallo::parse_problem_file(filename, spec)) return -1;
    auto has = [&](const char* k) { auto it = spec.constants.find(k); return it != spec.constants.end() && it->second > 0; };
    return has("materialize_JtJ") ? 2 : has("materialize_J") ? 1 : 0;
}

// Does the front-end derive, from this file, exactly the kernels it derives from one of the bundled files of `energy`?  (A file that matches a
// hand-written plugin by its declarations only -- same inputs, same residual names -- may still state a different energy.)
namespace thallo {
bool unit_matches_bundled(const char* filename, const std::string& energy)
{
    std::string err;
    const unsigned long long h = dsl::unit_fingerprint(filename, err);
    if (!h) return false;
    for (const auto& k : KNOWN_UNIT_HASHES) if (k.hash == h && energy == k.energy) return true;
    return false;
}
}  // namespace thallo

extern "C" unsigned long long ThalloX_ProblemFileUnitHash(const char* filename)
{
    std::string err;
    const unsigned long long h = filename ? thallo::dsl::unit_fingerprint(filename, err) : 0ULL;
    if (!h) thallo::set_error("%s", err.c_str());
    return h;
}

extern "C" unsigned long long ThalloX_ProblemFileHash(const char* filename, char* energy_out, int cap)
{
    thallo::ProblemSpec spec;
    thallo::parse_problem_file(filename, spec);
    if (energy_out && cap > 0) { snprintf(energy_out, cap, "%s", spec.energy.c_str()); }
    return spec.body_hash;
}
