// frontend_fuzz.cpp -- the mini front-end (csrc/dsl_lua.cpp, csrc/dsl_codegen.cpp) under AddressSanitizer / UBSan on the CPU, without a device:
// every .t given on the command line is run and lowered as it is, then MUTATED (truncations, byte flips, spliced lines, duplicated tokens, huge numbers; a fixed
// seed) -- a mutant must either lower or be refused with a message; a crash, a sanitizer report or a run-away is the failure.  An energy file is input
// from outside the library (Thallo_ProblemDefine takes a path), so the interpreter has to survive anything.  tests/test_frontend_sanitized.py builds and runs it:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all tools/frontend_fuzz.cpp thallo_amd/csrc/dsl_lua.cpp thallo_amd/csrc/dsl_codegen.cpp
#include "../thallo_amd/csrc/dsl.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unistd.h>

namespace thallo { const char* env_switch(const char*) { return nullptr; } }      // (solver.cpp's table; the front-end asks for THALLO_FRONTEND_AGGREGATE)
#ifdef WITH_RECOGNISER      // + csrc/frontend.cpp (the recogniser of the bundled energies: its own lexer, declaration parser and hashes), built with -DWITH_RECOGNISER -I/opt/rocm/include -D__HIP_PLATFORM_AMD__
#include <cstdarg>
namespace thallo {
static char g_err[4096];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
const char* last_error() { return g_err; }
struct ProblemSpec;
}
extern "C" int ThalloX_ProblemFileSchedule(const char* filename);
extern "C" unsigned long long ThalloX_ProblemFileUnitHash(const char* filename);
extern "C" unsigned long long ThalloX_ProblemFileHash(const char* filename, char* energy_out, int cap);
static void recognise(const char* path)
{
    char energy[64];
    (void)ThalloX_ProblemFileHash(path, energy, (int)sizeof energy);
    (void)ThalloX_ProblemFileSchedule(path);
}
#else
static void recognise(const char*) {}
#endif

static unsigned long long rng_state = 0x9e3779b97f4a7c15ULL;
static unsigned rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 32); }

static int lower(const std::string& path, const unsigned* dims, std::string* why)
{
    recognise(path.c_str());
    thallo::dsl::Problem p; std::string err;
    if (!thallo::dsl::run_problem_file(path.c_str(), p, err, dims)) { if (why) *why = err; return 1; }
    for (int f64 = 0; f64 < 2; ++f64) {
        thallo::dsl::Generated g;
        if (!thallo::dsl::generate_source(p, g, err, f64 != 0)) { if (why) *why = err; return 2; }
        if (g.source.empty()) { if (why) *why = "empty translation unit"; return 3; }
    }
    (void)thallo::dsl::describe(p);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: frontend_fuzz <mutants per file> <file.t> ...\n"); return 2; }
    const int mutants = atoi(argv[1]);
    if (const char* seed = getenv("THALLO_FUZZ_SEED")) rng_state ^= strtoull(seed, nullptr, 0) * 0x9e3779b97f4a7c15ULL;      // (another stream of mutants; the tests use the built-in seed)
    if (!rng_state) rng_state = 0x2545f4914f6cdd1dULL;
    const unsigned dims[16] = { 7, 5, 6, 4, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2 };       // small sizes: files with Sum are expanded for them
    char tmpl[] = "/tmp/thallo_fuzz_XXXXXX";
    const int fd = mkstemp(tmpl); if (fd < 0) { perror("mkstemp"); return 2; }
    close(fd);
    long ok = 0, refused = 0, intact_ok = 0;
    for (int a = 2; a < argc; ++a) {
        std::ifstream in(argv[a], std::ios::binary); std::stringstream ss; ss << in.rdbuf(); const std::string text = ss.str();
        std::string why;
        const int rc = lower(argv[a], dims, &why);
        if (rc == 0) ++intact_ok; else printf("%s: refused as it is: %s\n", argv[a], why.c_str());
        for (int m = 0; m < mutants; ++m) {
            std::string t = text;
            const int n_edits = 1 + (int)(rnd() % 3);
            for (int e = 0; e < n_edits && !t.empty(); ++e) {
                const size_t pos = rnd() % t.size();
                switch (rnd() % 8) {
                    case 0: t.resize(pos); break;                                                                   // truncated
                    case 1: t[pos] = (char)(rnd() & 0xff); break;                                                   // a byte flipped
                    case 2: t.erase(pos, 1 + rnd() % 12); break;                                                    // something missing
                    case 3: { const size_t from = rnd() % t.size(); t.insert(pos, t.substr(from, 1 + rnd() % 40)); break; }      // spliced
                    case 4: t.insert(pos, "99999999999999999999"); break;                                           // a huge number
                    case 5: t.insert(pos, "(((((((((((((((((((((((((((((((("); break;                                 // deep nesting
                    case 6: t.insert(pos, " -1 "); break;
                    default: { const char* kw[] = { " end ", " function ", " local ", " Sum(", " {", "}", ")", " for i = 1, 1e9 do ", "..", " X(x, y, z, w) " }; t.insert(pos, kw[rnd() % 10]); break; }
                }
            }
            { std::ofstream out(tmpl, std::ios::binary); out << t; }
            if (lower(tmpl, dims, nullptr) == 0) ++ok; else ++refused;
        }
    }
    unlink(tmpl);
    printf("files %d (%ld lower as they are), mutants %ld lowered, %ld refused with a message\n", argc - 2, intact_ok, ok, refused);
    return 0;
}
