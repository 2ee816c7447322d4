// formats_fuzz.cpp -- the harness's file readers (harness/data_formats.hpp, harness/image_io.hpp: PNG, .constraints, .imagedump, the 160-byte SFS parameter
// struct, OFF / PLY meshes, .mrk landmarks) under AddressSanitizer / UBSan on the CPU: each seed file as it is, then seeded mutants (truncations, flipped bytes,
// spliced ranges, huge counts).  A reader must return or throw std::runtime_error -- a sanitizer report, std::bad_alloc from a count the file cannot back,
// or a run-away loop is the failure.  tests/test_formats_sanitized.py builds and runs it (g++ ... -lz).
//   usage: formats_fuzz <mutants per file> <kind>:<file> ...      kind = png | constraints | imagedump | sfsparams | off | ply | mrk
#include "../harness/data_formats.hpp"
#include "../harness/image_io.hpp"
#include <cstdio>
#include <cstdlib>
#include <new>
#include <unistd.h>

static unsigned long long rng_state = 0x243f6a8885a308d3ULL;
static unsigned rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 32); }

static int read_one(const std::string& kind, const std::string& path)
{
    try {
        if (kind == "png") { auto im = harness::read_png(path); volatile size_t n = im.px.size(); (void)n; }
        else if (kind == "constraints") { auto c = harness::read_constraints(path); volatile size_t n = c.size(); (void)n; }
        else if (kind == "imagedump") { auto d = harness::read_imagedump(path); volatile size_t n = d.f.size() + d.u.size(); (void)n; }
        else if (kind == "sfsparams") { auto p = harness::read_sfs_parameters(path); volatile float f = p.fx; (void)f; }
        else if (kind == "off" || kind == "ply") {
            auto m = kind == "off" ? harness::read_off(path) : harness::read_ply(path);
            std::vector<int> v0, v1; harness::directed_edges(m, v0, v1);
        }
        else if (kind == "mrk") { auto l = harness::read_mrk(path); volatile size_t n = l.index.size(); (void)n; }
        else { fprintf(stderr, "unknown kind %s\n", kind.c_str()); exit(2); }
    } catch (const std::bad_alloc&) { fprintf(stderr, "%s reader: std::bad_alloc (a count was trusted)\n", kind.c_str()); exit(3); }
    catch (const std::length_error&) { fprintf(stderr, "%s reader: std::length_error (a count was trusted)\n", kind.c_str()); exit(3); }
    catch (const std::exception&) { return 1; }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: formats_fuzz <mutants per file> <kind>:<file> ...\n"); return 2; }
    const int mutants = atoi(argv[1]);
    if (const char* seed = getenv("THALLO_FUZZ_SEED")) rng_state ^= strtoull(seed, nullptr, 0) * 0x9e3779b97f4a7c15ULL;      // (another stream of mutants; the tests use the built-in seed)
    if (!rng_state) rng_state = 0x2545f4914f6cdd1dULL;
    long ok = 0, refused = 0, intact = 0;
    for (int a = 2; a < argc; ++a) {
        const std::string arg = argv[a]; const size_t colon = arg.find(':');
        const std::string kind = arg.substr(0, colon), path = arg.substr(colon + 1);
        std::ifstream in(path, std::ios::binary); std::stringstream ss; ss << in.rdbuf(); const std::string data = ss.str();
        if (read_one(kind, path) == 0) ++intact; else printf("%s: refused as it is\n", path.c_str());
        char tmpl[64]; snprintf(tmpl, sizeof tmpl, "/tmp/thallo_fmt_XXXXXX"); const int fd = mkstemp(tmpl); if (fd < 0) return 2; close(fd);
        const std::string tmp = std::string(tmpl) + (kind == "off" ? ".off" : "");
        for (int m = 0; m < mutants; ++m) {
            std::string t = data;
            const int edits = 1 + (int)(rnd() % 3);
            for (int e = 0; e < edits && !t.empty(); ++e) {
                const size_t pos = rnd() % t.size();
                switch (rnd() % 7) {
                    case 0: t.resize(pos); break;
                    case 1: t[pos] = (char)(rnd() & 0xff); break;
                    case 2: t.erase(pos, 1 + rnd() % 16); break;
                    case 3: { const size_t from = rnd() % t.size(); t.insert(pos, t.substr(from, 1 + rnd() % 64)); break; }
                    case 4: t.insert(pos, "999999999999"); break;
                    case 5: { for (int k = 0; k < 4 && pos + k < t.size(); ++k) t[pos + k] = (char)0xff; break; }
                    default: t.insert(pos, " -7 "); break;
                }
            }
            { std::ofstream out(tmp, std::ios::binary); out << t; }
            if (read_one(kind, tmp) == 0) ++ok; else ++refused;
        }
        unlink(tmp.c_str()); if (tmp != tmpl) unlink(tmpl);
    }
    printf("files %d (%ld read as they are), mutants %ld read, %ld refused with a message\n", argc - 2, intact, ok, refused);
    return 0;
}
