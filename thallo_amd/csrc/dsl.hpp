// dsl.hpp -- the mini front-end for `.t` problem specifications.
//
// The reference executes a `.t` file in a Lua/Terra sandbox whose globals are the DSL (API/src/thallo.t:1359-1434, 1580-2112;
// API/src/lib.t:18-594): the file's operators build an expression graph, API/src/ad.t differentiates it symbolically and
// thallo.t:3536-3949 emits residual-wise kernels through Terra -> PTX.  Here:
//   dsl_lua.cpp     a small interpreter for the Lua subset the energy files use (locals, tables, functions / closures, for-in over
//                   Stencil{}, method calls), with the DSL's constructors and the lib.t helpers as builtins; running a file yields a
//                   `Problem`: declarations + one scalar expression DAG per residual component;
//   dsl_codegen.cpp emits ONE HIP translation unit per problem: per named residual a cost / evalJTF / applyJTJ (and applyJ / applyJt)
//                   kernel in the reference's residual-wise form (thallo.t:3536-3569, 3867-3908, 3939-3949).  Derivatives come from
//                   forward-mode dual numbers over the residual's unknown accesses -- the rules of ad.t:698-836 applied at run time
//                   instead of symbolically (Select: partials (0, c, not c), comparisons: zero derivative);
//   dsl_plugin.cpp  compiles that unit with hipRTC for gfx950 at Plan time and drives it through the EnergyPlugin interface
//                   (unfused PCG schedule; scatter-adds are float atomics, exactly the reference's residual-wise lowering).
// Hand-written plugins stay the fast path for the bundled energies; a file they do not recognise -- or any file under
// THALLO_FRONTEND=generate -- goes through here.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace thallo {
namespace dsl {

enum class Op {
    Const, Param, Load, IndexVal,                  // leaves: literal, scalar Param, image access, iteration index as a value (x:asvalue())
    Add, Sub, Mul, Div, Neg, Sqrt, Sin, Cos, Abs, Pow,
    Sample,                                        // SampledImage(im, dx, dy)(x, y): bilinear sample of an Array at the float position (a[0], a[1]); partials are the
                                                   // samples of the two derivative images (thallo.t:5798-5817, Image:sample thallo.t:899-907)
    Detach,                                        // lib.t Constant(e) = ad.constant: the value of e with no derivative (robust-norm weights, lib.t:157-169)
    Select,                                        // a[0] != 0 ? a[1] : a[2]
    Eq, Ge, Gt, Le, Lt, Not, And, Or,              // 0 / 1 valued, zero derivative (ad.t:824-829)
    InBounds,                                      // all index components inside the iteration-dimension bounds (thallo.t:1993-1997)
    Computed                                       // round 6: expr:get(x + dx, y + dy) -- component `channel` of computed array `input` (Problem::computed) at the shift idx; a[0] is the
                                                   // same access INLINED (Select(InBounds, the shifted expression, 0)): what runs where the array is not materialized
};

// one component of an image index: iteration variable of dimension `dim` plus a constant offset, optionally through a Sparse map
// (graph domains: X(v0(e)): dim = the edge dimension, sparse = input slot of v0; the offset applies before the map and is 0 there)
// A Sparse map over a 2-D domain (Sparse({W,H},{W},k): Xn(x, y)) is looked up with TWO iteration variables: dim2 = the second one (-1: a 1-D map).
// Round 3 (Sum / tensor contraction, lib.t:146): plain components are affine in up to TWO iteration variables, sign * var(dim) + sign_b * var(dim_b) + off
// (convolution.t: R(n - k + 2)), and a component with no variable left (dim < 0, sparse < 0) is the constant index `off` -- what a summed variable becomes
// when Sum is expanded (W(m) -> W(0), W(1), ...).
struct IndexComp { int dim = -1; int off = 0; int sparse = -1; int dim2 = -1; int sign = 1; int dim_b = -1; int sign_b = 1; };
inline bool operator==(const IndexComp& a, const IndexComp& b)
{ return a.dim == b.dim && a.off == b.off && a.sparse == b.sparse && a.dim2 == b.dim2 && a.sign == b.sign && a.dim_b == b.dim_b && a.sign_b == b.sign_b; }
inline bool plain_index(const IndexComp& a) { return a.dim >= 0 && a.sparse < 0 && a.dim2 < 0 && a.sign == 1 && a.dim_b < 0; }      // var + off: what rounds 1-2 knew

struct Expr;
typedef std::shared_ptr<const Expr> E;
struct Expr {
    Op op = Op::Const;
    double c = 0.0;                                // Const
    int input = -1;                                // Param / Load: index into Problem::inputs
    int channel = 0;                               // Load / Sample
    int input_dx = -1, input_dy = -1;              // Sample: the derivative images (-1: none given, the sample then must not depend on an unknown)
    std::vector<IndexComp> idx;                    // Load / InBounds ; IndexVal: idx[0]
    int expand = 0;                                // InBounds: InBoundsExpanded(x, y, n) keeps n pixels off the border
    std::vector<E> a;                              // operands
};

enum class InputKind { Unknown, Array, Sparse, Param };
struct Input {
    std::string name;
    InputKind kind = InputKind::Array;
    int channels = 1;
    bool is_u8 = false;                            // Array(uint8, ...)
    bool fixed_f64 = false;                        // declared `double`: double whatever the precision of the state -- supported under doublePrecision = 1 only
    bool fixed_f32 = false;                        // declared with a fixed single-precision element type (float, float3, mat3f ...) rather than a thallo_float one:
                                                   // stays float under doublePrecision = 1 (precision.t:3-6 switches thallo_float only)
    std::vector<int> dims;                         // dimension ids (Unknown / Array: the image's; Sparse: {from..., to}: one or two source dimensions, then the target)
    int slot = -1;                                 // index into the void** problem parameters
    E exclude;                                     // Unknown:Exclude(cond), evaluated at the unknown's own index (may be null)
};

struct Residual {
    std::string name;
    std::vector<E> exprs;                          // scalar components
    std::vector<int> domain;                       // iteration dimensions (ids), in first-use order
    bool mat_J = false, mat_JtJ = false, mat_Jp = false;   // r.<name>.J / JtJ / Jp :set_materialize(true) (thallo.t:5757-5772)
    int at_output = -1;                            // r.<name>:compute_at_output(b) (thallo.t:5661-5674): 1 = unknown-wise (gather) lowering asked for, 0 = residual-wise, -1 = not said
};

// A computed array (thallo.t:1868-1937): an expression over an iteration domain that the energy reads back through :get at shifted positions.  The reference materializes it
// -- one image for the value and one GRADIENT image per unknown the expression depends on -- with a `precompute` kernel per Gauss-Newton iteration (and again after an LM
// revert; gauss_newton.t:979-986,1748; thallo.t:4046-4094) instead of re-evaluating the expression at every access and in every PCG iteration.
struct ComputedArray { std::vector<E> exprs; std::vector<int> domain; };

struct Problem {
    std::string file;
    std::vector<std::string> dims;                 // Dims("W","H"): ids are positions; sizes come from the unsigned[] at Plan time
    std::vector<int> dim_alias;                    // per id: -1 = a declared dimension; d = a further ITERATION VARIABLE over dimension d (the second, third ... call of
                                                   // Kd(): thallo.t:467-477 makes a new IndexDomain per call) -- same size, behind the declared ones, not part of the caller's array
    int canonical(int d) const { return d >= 0 && d < (int)dim_alias.size() && dim_alias[d] >= 0 ? dim_alias[d] : d; }
    std::vector<Input> inputs;                     // in Inputs{} order (= declaration order of the unknown images in the flat vectors)
    bool use_preconditioner = false;
    bool direct_solve = false;                     // <Residuals handle>:set_direct_solve(true) (thallo.t:5634-5636); acted on only under THALLO_ENABLE_DIRECT_SOLVE=1,
                                                   // like the reference's compile-time enable_direct_solve (gauss_newton.t:22)
    std::vector<Residual> residuals;
    std::vector<ComputedArray> computed;           // the expressions read through :get with a pure shift (Op::Computed nodes refer to them by index)
    int max_slot = -1;
    const unsigned* plan_dims = nullptr;           // Thallo_ProblemPlan's dimensions while the file runs (NULL: not known; Sum then is an error); read per declared dimension
    std::vector<long> dim_sizes;                   // ... the sizes of the dimensions declared so far (-1: unknown)
};

// dsl_lua.cpp: run a .t file.  false + `err` on anything outside the supported subset (never a silent partial result).
bool run_problem_file(const char* filename, Problem& out, std::string& err, const unsigned* dims = nullptr);

// dsl_codegen.cpp
struct GenKernel { std::string name; int residual; int kind; };      // kind: 0 cost, 1 evalJTF, 2 applyJTJ, 3 applyJ (Jp = J p), 4 applyJt (Ap += J^T Jp), 5 dumpJ (materialize the rows),
                                                                      // 6 evalJTF / 7 applyJTJ in the unknown-wise (gather) form -- present only where gather_ok
constexpr int GEN_KINDS = 8;
// one merged gather kernel pair per iteration domain (dsl_codegen.cpp): the member residuals, the (input, channel) targets they write, the kernel names
struct GenGroup { std::vector<int> domain; std::vector<int> members; std::vector<std::pair<int, int>> targets; std::string jtj, jtf, cost; };      // cost: "" = the members' own cost kernels
// Unknown-wise lowering THROUGH index maps (round 4; dsl_codegen.cpp "incidence gather"): residuals that reach their unknowns through Sparse maps (or any other index
// the stencil gather cannot invert at code-generation time).  The unknown images are grouped by their dimension lists (an "owner" = one pixel of that index space, all
// images over it and all their channels); per (residual, group) a kernel pair walks the owners, each thread evaluating the residual instances in ITS list -- built by the
// plugin per Init from the residual's own index evaluation (uidx kernel) -- and keeping only the partials of its own unknowns: one writer per element, no atomics.
struct IncGroup { std::vector<int> dims; std::vector<int> inputs; };
struct IncResidual { int ri = -1, K = 0; std::vector<int> slot_input; std::string uidx; std::vector<int> groups; std::vector<std::string> jtj, jtf; };
// a materialized computed array: its precompute kernel and its planes in Ctx::cap -- per component the value plane, then one plane per unknown access with a structurally
// non-zero partial (planes of one array are consecutive; every plane has one float per element of the array's domain)
struct GenComputed { std::string kernel; std::vector<int> domain; int plane0 = 0, planes = 0; };
struct Generated {
    std::vector<GenComputed> computed;             // (only the arrays that are materialized: at most 48 unknown accesses, materialization not switched off)
    int n_planes = 0;
    std::vector<IncGroup> inc_groups;
    std::vector<IncResidual> inc;                  // (residuals without the stencil gather, at most 48 unknown accesses)
    std::string source;                            // one HIP translation unit
    std::vector<GenKernel> kernels;
    std::vector<int> slots_per_row;                // per residual: K, the entries per materialized row
    std::vector<char> gather_ok;                   // per residual: the unknown-wise lowering exists (residual dims == the dims of every unknown it reads, constant-offset stencil accesses)
    std::vector<char> want_gather;                 // IN (optional): per residual, the caller wants it gathered -- only those join a merged group kernel (empty: every eligible one)
    std::vector<GenGroup> groups;                  // the merged gather kernels, one pair per iteration domain
    std::vector<long> jp_offset;                   // per residual: offset of its rows in the Jp vector (Jt[Jp] schedule), in units of elements x components
    int n_prm = 0;
    bool has_wide = false;                         // some residual took the wide lowering (more than 48 unknown accesses): the plugin compiles the unit without loop unrolling
};
// f64 (Thallo_InitializationParameters::doublePrecision): constants as double literals; the plugin compiles the unit with `float` standing for double
bool generate_source(const Problem& p, Generated& out, std::string& err, bool f64 = false);

// FNV-1a-64 of the translation unit generated from a .t file, residual names aside: two files with the same fingerprint state the same energy
// (same expression DAG per residual component, same unknown accesses, same guards).  0 + err if the file is outside the supported subset.
unsigned long long unit_fingerprint(const char* filename, std::string& err);

const char* generated_prelude();                   // the fixed text every generated unit starts with (Dual, helpers, the wave64 primitives)
std::string describe(const Problem& p);            // one-line-per-declaration summary (tests / verbosity)

}  // namespace dsl
}  // namespace thallo
