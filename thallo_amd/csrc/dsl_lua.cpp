// dsl_lua.cpp -- interpreter for the Lua subset the Thallo energy files use; see dsl.hpp.
//
// What the reference's sandbox provides and this file restates as builtins (all citations relative to /root/reference/API/src):
//   Dims / Inputs / Unknown / Array / Sparse / Param / UsePreconditioner / Residuals      thallo.t:1580-2112, 5541-5631
//   image(ix, iy), sparse(e), dim(), x:asvalue(), Unknown:Exclude(c), expr:get(ix, iy)      thallo.t:1993-1997, 2091-2112, 2925-2943
//   Select, InBounds, InBoundsExpanded, eq / greater / greatereq / less / lesseq, Not, All, Any, abs, sqrt, sin, cos, Vector, dot,
//   cross, Rotate2D, Rotate3D, AngleAxisRotatePoint, Stencil, vec:slice(a, b), vec(i)       lib.t:18-594 (Rotate3D :123-137,
//                                                                                          Rotate2D :138-142, AngleAxisRotatePoint :514-555)
//   r.<name>.J / JtJ / Jp :set_materialize(b), r:merge(a, b), :compute_at_output(b)         thallo.t:5661-5772
// Lua itself: locals and globals, multiple assignment and multiple returns, tables (constructors, #t, t[k], t.k), functions and
// closures, numeric for, for-in over Stencil{} / ipairs / pairs, if / elseif / else, while, return, break, method calls, string and
// number literals, comments.  Not supported (-> an error naming the construct): metatables, coroutines, goto, varargs, string library.
#include "dsl.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <set>
#include <algorithm>
#include <sstream>
#include <stdexcept>

namespace thallo {
namespace dsl {

namespace {

[[noreturn]] void fail(const std::string& m) { throw std::runtime_error(m); }
// a number of the file that becomes an index offset, a vector component or a parameter slot: integral and small, or the file is refused (the cast of an
// out-of-range double, like the overflow of an offset sum, is undefined behaviour -- tools/frontend_fuzz.cpp found both)
constexpr long SMALL_INT_LIMIT = 1L << 20;
int small_int(double v, const std::string& what)
{
    if (!(v == std::floor(v)) || v < -(double)SMALL_INT_LIMIT || v > (double)SMALL_INT_LIMIT) fail(what + ": " + std::to_string(v) + " is not a small integer");
    return (int)v;
}
void add_off(int& off, long long delta)
{
    const long long s = (long long)off + delta;
    if (s < -SMALL_INT_LIMIT || s > SMALL_INT_LIMIT) fail("index offset out of range");
    off = (int)s;
}

// ================================================================================================ lexer
struct Tok { enum K { Name, Num, Str, Op, Kw, End } k = End; std::string s; double n = 0; int line = 0; };

const char* KEYWORDS[] = { "and", "break", "do", "else", "elseif", "end", "false", "for", "function", "if", "in", "local", "nil", "not", "or",
                           "repeat", "return", "then", "true", "until", "while", nullptr };

std::vector<Tok> lex(const std::string& src)
{
    std::vector<Tok> out; size_t i = 0; int line = 1;
    if (src.size() >= 3 && (unsigned char)src[0] == 0xEF && (unsigned char)src[1] == 0xBB && (unsigned char)src[2] == 0xBF) i = 3;
    auto long_bracket = [&](size_t j, std::string* text) -> size_t {     // src[j] == '[': returns index after the closing bracket, or 0
        size_t k = j + 1; int eq = 0;
        while (k < src.size() && src[k] == '=') { ++eq; ++k; }
        if (k >= src.size() || src[k] != '[') return 0;
        const std::string close = "]" + std::string(eq, '=') + "]";
        const size_t e = src.find(close, k + 1);
        if (e == std::string::npos) fail("unterminated long bracket");
        for (size_t q = k + 1; q < e; ++q) if (src[q] == '\n') ++line;
        if (text) *text = src.substr(k + 1, e - k - 1);
        return e + close.size();
    };
    while (i < src.size()) {
        const unsigned char c = src[i];
        if (c == '\n') { ++line; ++i; continue; }
        if (isspace(c)) { ++i; continue; }
        if (c == '-' && i + 1 < src.size() && src[i + 1] == '-') {
            if (i + 2 < src.size() && src[i + 2] == '[') { const size_t e = long_bracket(i + 2, nullptr); if (e) { i = e; continue; } }
            while (i < src.size() && src[i] != '\n') ++i;
            continue;
        }
        Tok t; t.line = line;
        if (isalpha(c) || c == '_') {
            size_t j = i; while (j < src.size() && (isalnum((unsigned char)src[j]) || src[j] == '_')) ++j;
            t.s = src.substr(i, j - i); t.k = Tok::Name;
            for (int q = 0; KEYWORDS[q]; ++q) if (t.s == KEYWORDS[q]) t.k = Tok::Kw;
            i = j;
        } else if (isdigit(c) || (c == '.' && i + 1 < src.size() && isdigit((unsigned char)src[i + 1]))) {
            char* endp = nullptr; t.n = strtod(src.c_str() + i, &endp); t.k = Tok::Num;
            size_t j = endp - src.c_str();
            if (j < src.size() && (src[j] == 'f' || src[j] == 'F')) ++j;      // tolerate a C-style suffix
            t.s = src.substr(i, j - i); i = j;
        } else if (c == '"' || c == '\'') {
            size_t j = i + 1; std::string v;
            while (j < src.size() && src[j] != (char)c) { if (src[j] == '\\' && j + 1 < src.size()) { ++j; v += src[j] == 'n' ? '\n' : src[j]; } else v += src[j]; ++j; }
            if (j >= src.size()) fail("unterminated string at line " + std::to_string(line));
            t.k = Tok::Str; t.s = v; i = j + 1;
        } else if (c == '[' && i + 1 < src.size() && (src[i + 1] == '[' || src[i + 1] == '=')) {
            std::string v; const size_t e = long_bracket(i, &v);
            if (e) { t.k = Tok::Str; t.s = v; i = e; } else { t.k = Tok::Op; t.s = "["; ++i; }
        } else {
            static const char* OPS3[] = { "...", nullptr };
            static const char* OPS2[] = { "==", "~=", "<=", ">=", "..", "::", nullptr };
            t.k = Tok::Op; t.s = std::string(1, (char)c);
            bool done = false;
            for (int q = 0; OPS3[q] && !done; ++q) if (src.compare(i, 3, OPS3[q]) == 0) { t.s = OPS3[q]; i += 3; done = true; }
            for (int q = 0; OPS2[q] && !done; ++q) if (src.compare(i, 2, OPS2[q]) == 0) { t.s = OPS2[q]; i += 2; done = true; }
            if (!done) ++i;
        }
        out.push_back(t);
    }
    Tok e; e.k = Tok::End; e.line = line; out.push_back(e);
    return out;
}

// ================================================================================================ AST
struct Node;
typedef std::shared_ptr<Node> NP;
struct Node {
    enum K { Nil, True, False, Number, String, Function, Name, Index, Call, Method, Binop, Unop, Table, Paren,
             Local, Assign, CallStat, Do, While, NumFor, GenFor, If, FuncStat, Return, Break, Block } k;
    int line = 0;
    double num = 0; std::string s;                 // literal / name / operator / method name
    std::vector<NP> a;                             // children (meaning per kind, see the parser)
    std::vector<std::string> names;                // Local / GenFor / NumFor variable names ; Function parameter names
    std::vector<NP> keys;                          // Table: key expression per field (null = positional)
    bool is_local = false;
    explicit Node(K kk, int ln) : k(kk), line(ln) {}
};

struct Parser {
    std::vector<Tok> t; size_t p = 0;
    explicit Parser(std::vector<Tok> toks) : t(std::move(toks)) {}
    const Tok& cur() const { return t[p]; }
    bool is_op(const char* s) const { return cur().k == Tok::Op && cur().s == s; }
    bool is_kw(const char* s) const { return cur().k == Tok::Kw && cur().s == s; }
    [[noreturn]] void err(const std::string& m) const { fail("line " + std::to_string(cur().line) + ": " + m + " near '" + cur().s + "'"); }
    void expect_op(const char* s) { if (!is_op(s)) err(std::string("expected '") + s + "'"); ++p; }
    void expect_kw(const char* s) { if (!is_kw(s)) err(std::string("expected '") + s + "'"); ++p; }
    std::string expect_name() { if (cur().k != Tok::Name) err("expected a name"); return t[p++].s; }

    NP block()
    {
        NP b = std::make_shared<Node>(Node::Block, cur().line);
        for (;;) {
            if (cur().k == Tok::End || is_kw("end") || is_kw("else") || is_kw("elseif") || is_kw("until")) break;
            if (is_op(";")) { ++p; continue; }
            NP s = statement();
            b->a.push_back(s);
            if (s->k == Node::Return || s->k == Node::Break) { if (is_op(";")) ++p; break; }
        }
        return b;
    }
    NP funcbody(int line)
    {
        NP f = std::make_shared<Node>(Node::Function, line);
        expect_op("(");
        if (!is_op(")")) for (;;) { if (is_op("...")) err("varargs are not supported"); f->names.push_back(expect_name()); if (is_op(",")) { ++p; continue; } break; }
        expect_op(")");
        f->a.push_back(block());
        expect_kw("end");
        return f;
    }
    std::vector<NP> exprlist() { std::vector<NP> v; v.push_back(expr()); while (is_op(",")) { ++p; v.push_back(expr()); } return v; }

    NP statement()
    {
        const int line = cur().line;
        if (is_kw("local")) {
            ++p;
            if (is_kw("function")) {
                ++p; NP s = std::make_shared<Node>(Node::FuncStat, line); s->is_local = true;
                NP nm = std::make_shared<Node>(Node::Name, line); nm->s = expect_name();
                s->a.push_back(nm); s->a.push_back(funcbody(line)); return s;
            }
            NP s = std::make_shared<Node>(Node::Local, line);
            s->names.push_back(expect_name());
            while (is_op(",")) { ++p; s->names.push_back(expect_name()); }
            if (is_op("=")) { ++p; s->a = exprlist(); }
            return s;
        }
        if (is_kw("function")) {
            ++p; NP s = std::make_shared<Node>(Node::FuncStat, line);
            NP target = std::make_shared<Node>(Node::Name, line); target->s = expect_name();
            while (is_op(".")) { ++p; NP ix = std::make_shared<Node>(Node::Index, line); NP key = std::make_shared<Node>(Node::String, line); key->s = expect_name(); ix->a = { target, key }; target = ix; }
            if (is_op(":")) err("method definitions are not supported");
            s->a.push_back(target); s->a.push_back(funcbody(line)); return s;
        }
        if (is_kw("if")) {
            ++p; NP s = std::make_shared<Node>(Node::If, line);
            s->a.push_back(expr()); expect_kw("then"); s->a.push_back(block());
            while (is_kw("elseif")) { ++p; s->a.push_back(expr()); expect_kw("then"); s->a.push_back(block()); }
            if (is_kw("else")) { ++p; s->a.push_back(nullptr); s->a.push_back(block()); }
            expect_kw("end"); return s;
        }
        if (is_kw("while")) { ++p; NP s = std::make_shared<Node>(Node::While, line); s->a.push_back(expr()); expect_kw("do"); s->a.push_back(block()); expect_kw("end"); return s; }
        if (is_kw("do")) { ++p; NP s = std::make_shared<Node>(Node::Do, line); s->a.push_back(block()); expect_kw("end"); return s; }
        if (is_kw("for")) {
            ++p; std::vector<std::string> names; names.push_back(expect_name());
            if (is_op("=")) {
                ++p; NP s = std::make_shared<Node>(Node::NumFor, line); s->names = names;
                s->a.push_back(expr()); expect_op(","); s->a.push_back(expr());
                if (is_op(",")) { ++p; s->a.push_back(expr()); } else s->a.push_back(nullptr);
                expect_kw("do"); s->a.push_back(block()); expect_kw("end"); return s;
            }
            while (is_op(",")) { ++p; names.push_back(expect_name()); }
            expect_kw("in");
            NP s = std::make_shared<Node>(Node::GenFor, line); s->names = names;
            s->a = exprlist();
            expect_kw("do"); NP body = block(); expect_kw("end");
            s->a.push_back(body); return s;
        }
        if (is_kw("return")) {
            ++p; NP s = std::make_shared<Node>(Node::Return, line);
            if (!(cur().k == Tok::End || is_kw("end") || is_kw("else") || is_kw("elseif") || is_kw("until") || is_op(";"))) s->a = exprlist();
            return s;
        }
        if (is_kw("break")) { ++p; return std::make_shared<Node>(Node::Break, line); }
        if (is_kw("repeat") || is_kw("goto")) err("statement not supported");
        // assignment or call
        NP e = suffixedexp();
        if (is_op("=") || is_op(",")) {
            NP s = std::make_shared<Node>(Node::Assign, line);
            std::vector<NP> targets{ e };
            while (is_op(",")) { ++p; targets.push_back(suffixedexp()); }
            expect_op("=");
            std::vector<NP> vals = exprlist();
            s->keys = targets; s->a = vals;
            for (auto& tg : targets) if (tg->k != Node::Name && tg->k != Node::Index) err("cannot assign to this expression");
            return s;
        }
        if (e->k != Node::Call && e->k != Node::Method) err("syntax error (expression is not a statement)");
        NP s = std::make_shared<Node>(Node::CallStat, line); s->a.push_back(e); return s;
    }

    NP primaryexp()
    {
        const int line = cur().line;
        if (cur().k == Tok::Name) { NP n = std::make_shared<Node>(Node::Name, line); n->s = t[p++].s; return n; }
        if (is_op("(")) { ++p; NP e = expr(); expect_op(")"); NP pr = std::make_shared<Node>(Node::Paren, line); pr->a.push_back(e); return pr; }
        err("unexpected symbol");
    }
    std::vector<NP> callargs()
    {
        std::vector<NP> args;
        if (cur().k == Tok::Str) { NP s = std::make_shared<Node>(Node::String, cur().line); s->s = t[p++].s; args.push_back(s); return args; }
        if (is_op("{")) { args.push_back(tablector()); return args; }
        expect_op("(");
        if (!is_op(")")) args = exprlist();
        expect_op(")");
        return args;
    }
    NP suffixedexp()
    {
        NP e = primaryexp();
        for (;;) {
            const int line = cur().line;
            if (is_op(".")) { ++p; NP ix = std::make_shared<Node>(Node::Index, line); NP key = std::make_shared<Node>(Node::String, line); key->s = expect_name(); ix->a = { e, key }; e = ix; }
            else if (is_op("[")) { ++p; NP ix = std::make_shared<Node>(Node::Index, line); NP key = expr(); expect_op("]"); ix->a = { e, key }; e = ix; }
            else if (is_op(":")) { ++p; NP m = std::make_shared<Node>(Node::Method, line); m->s = expect_name(); m->a.push_back(e); for (auto& x : callargs()) m->a.push_back(x); e = m; }
            else if (is_op("(") || is_op("{") || cur().k == Tok::Str) { NP c = std::make_shared<Node>(Node::Call, line); c->a.push_back(e); for (auto& x : callargs()) c->a.push_back(x); e = c; }
            else return e;
        }
    }
    NP tablector()
    {
        NP tb = std::make_shared<Node>(Node::Table, cur().line);
        expect_op("{");
        while (!is_op("}")) {
            if (cur().k == Tok::Name && t[p + 1].k == Tok::Op && t[p + 1].s == "=") {
                NP key = std::make_shared<Node>(Node::String, cur().line); key->s = t[p].s; p += 2;
                tb->keys.push_back(key); tb->a.push_back(expr());
            } else if (is_op("[")) { ++p; NP key = expr(); expect_op("]"); expect_op("="); tb->keys.push_back(key); tb->a.push_back(expr()); }
            else { tb->keys.push_back(nullptr); tb->a.push_back(expr()); }
            if (is_op(",") || is_op(";")) ++p; else break;
        }
        expect_op("}");
        return tb;
    }
    NP simpleexp()
    {
        const int line = cur().line;
        if (cur().k == Tok::Num) { NP n = std::make_shared<Node>(Node::Number, line); n->num = t[p++].n; return n; }
        if (cur().k == Tok::Str) { NP n = std::make_shared<Node>(Node::String, line); n->s = t[p++].s; return n; }
        if (is_kw("nil")) { ++p; return std::make_shared<Node>(Node::Nil, line); }
        if (is_kw("true")) { ++p; return std::make_shared<Node>(Node::True, line); }
        if (is_kw("false")) { ++p; return std::make_shared<Node>(Node::False, line); }
        if (is_op("{")) return tablector();
        if (is_kw("function")) { ++p; return funcbody(line); }
        if (is_op("...")) err("varargs are not supported");
        return suffixedexp();
    }
    static int binprec(const Tok& tk, int& right)
    {
        right = 0;
        if (tk.k == Tok::Kw) { if (tk.s == "or") return 1; if (tk.s == "and") return 2; return 0; }
        if (tk.k != Tok::Op) return 0;
        const std::string& s = tk.s;
        if (s == "<" || s == ">" || s == "<=" || s == ">=" || s == "~=" || s == "==") return 3;
        if (s == "..") { right = 1; return 5; }
        if (s == "+" || s == "-") return 6;
        if (s == "*" || s == "/" || s == "%") return 7;
        if (s == "^") { right = 1; return 10; }
        return 0;
    }
    NP subexpr(int limit)
    {
        NP left;
        const int line = cur().line;
        if (is_kw("not") || is_op("-") || is_op("#")) {
            NP u = std::make_shared<Node>(Node::Unop, line); u->s = t[p++].s;
            u->a.push_back(subexpr(8)); left = u;
        } else left = simpleexp();
        for (;;) {
            int right; const int prec = binprec(cur(), right);
            if (prec == 0 || prec <= limit) break;
            NP b = std::make_shared<Node>(Node::Binop, cur().line); b->s = t[p++].s;
            NP rhs = subexpr(right ? prec - 1 : prec);
            b->a = { left, rhs }; left = b;
        }
        return left;
    }
    NP expr() { return subexpr(0); }
};

// ================================================================================================ values
struct Value;
struct TableV { std::vector<Value> arr; std::vector<std::pair<std::string, Value>> fields; };       // arr: 1-based positional part; fields keep insertion order
struct Env;
struct FuncV { int builtin = -1; std::string bname; NP def; std::shared_ptr<Env> env; };
struct SymV {
    enum K { Scalar, Vec, Dim, IndexDomain, IndexE, Image, Sampled, TypeName, ResidualsH, NamedRes, MatInfo, StencilList } k = Scalar;
    E e; std::vector<E> v;                         // Scalar / Vec
    int id = -1;                                   // Dim / IndexDomain: dimension id ; Image / Sampled: input index ; NamedRes: residual index
    int id_dx = -1, id_dy = -1;                    // Sampled: the derivative images
    IndexComp ic;                                  // IndexE
    std::string s;                                 // TypeName ; MatInfo: "J" / "JtJ" / "Jp"
    std::vector<std::vector<double>> stencil;      // StencilList
};
struct Value {
    enum T { Nil, Bool, Num, Str, Table, Func, Sym } t = Nil;
    bool b = false; double n = 0; std::string s;
    std::shared_ptr<TableV> tab; std::shared_ptr<FuncV> fn; std::shared_ptr<SymV> sym;
    static Value num(double x) { Value v; v.t = Num; v.n = x; return v; }
    static Value boolean(bool x) { Value v; v.t = Bool; v.b = x; return v; }
    static Value str(const std::string& x) { Value v; v.t = Str; v.s = x; return v; }
    static Value make_sym(const SymV& sv) { Value v; v.t = Sym; v.sym = std::make_shared<SymV>(sv); return v; }
    bool truthy() const { return !(t == Nil || (t == Bool && !b)); }
};
typedef std::vector<Value> Values;

// A closure holds its defining environment and that environment holds the closure (`local function f ... end`): a shared_ptr cycle per function of the file, i.e. a
// leak per Thallo_ProblemPlan of any file that defines one (AddressSanitizer on tools/frontend_fuzz.cpp: image_warping.t, shape_from_shading.t ...).  Every Env of a
// run registers here; when the interpreter goes -- normally or through an exception -- the variables of all that are still alive are dropped, which opens every cycle.
struct Env;
static thread_local std::set<Env*>* g_live_envs = nullptr;
struct Env : std::enable_shared_from_this<Env> {
    Env() { if (g_live_envs) g_live_envs->insert(this); }
    ~Env() { if (g_live_envs) g_live_envs->erase(this); }
    Env(const Env&) = delete; Env& operator=(const Env&) = delete;
    std::map<std::string, Value> vars; std::shared_ptr<Env> parent;
    Value* find(const std::string& n) { for (Env* e = this; e; e = e->parent.get()) { auto it = e->vars.find(n); if (it != e->vars.end()) return &it->second; } return nullptr; }
};

// ================================================================================================ expression helpers
E mk(Op op, std::vector<E> a) { auto e = std::make_shared<Expr>(); e->op = op; e->a = std::move(a); return e; }
E konst(double c) { auto e = std::make_shared<Expr>(); e->op = Op::Const; e->c = c; return e; }
bool is_const(const E& e, double* c = nullptr) { if (e->op != Op::Const) return false; if (c) *c = e->c; return true; }
E bin(Op op, const E& a, const E& b)
{
    double x, y;
    if (is_const(a, &x) && is_const(b, &y)) {       // fold literals (the file's own arithmetic on numbers: weights, thresholds)
        switch (op) { case Op::Add: return konst(x + y); case Op::Sub: return konst(x - y); case Op::Mul: return konst(x * y); case Op::Div: return konst(x / y); default: break; }
    }
    // additive identities (x + 0, 0 + x, x - 0 are x exactly -- but for the sign of a zero; the reference's own simplifier drops them too, ad.t): what an
    // energy file writes as `offX + posX` with a literal 0 offset becomes the same node as `posX`
    if (op == Op::Add && is_const(a, &x) && x == 0.0) return b;
    if ((op == Op::Add || op == Op::Sub) && is_const(b, &y) && y == 0.0) return a;
    return mk(op, { a, b });
}
E un(Op op, const E& a)
{
    double x;
    if (is_const(a, &x)) switch (op) { case Op::Neg: return konst(-x); case Op::Sqrt: return konst(std::sqrt(x)); case Op::Sin: return konst(std::sin(x)); case Op::Cos: return konst(std::cos(x));
                                        case Op::Abs: return konst(std::fabs(x)); default: break; }
    return mk(op, { a });
}

// shift every index of an expression (expr:get(x+dx, y+dy)): per dimension id the extra offset
E shift_expr(const E& e, const std::map<int, int>& sh, std::map<const Expr*, E>& memo)
{
    auto it = memo.find(e.get()); if (it != memo.end()) return it->second;
    auto n = std::make_shared<Expr>(*e);
    for (auto& ic : n->idx) {
        auto f = sh.find(ic.dim); if (f != sh.end()) { if (ic.sparse >= 0 && f->second != 0) fail(":get with an offset through a Sparse map is not supported"); add_off(ic.off, (long long)ic.sign * f->second); }
        auto fb = sh.find(ic.dim_b); if (ic.dim_b >= 0 && fb != sh.end()) add_off(ic.off, (long long)ic.sign_b * fb->second);
    }
    for (auto& c : n->a) c = shift_expr(c, sh, memo);
    E r = n; memo[e.get()] = r; return r;
}
// expr:get(v0(e)): the expression's iteration variable over dimension `from` becomes the Sparse-mapped index (graph access to a computed
// expression, thallo.t:876-883 with a Sparse index): every plain occurrence of that variable is replaced
E subst_expr(const E& e, const std::map<int, IndexComp>& to, std::map<const Expr*, E>& memo)
{
    auto it = memo.find(e.get()); if (it != memo.end()) return it->second;
    auto n = std::make_shared<Expr>(*e);
    for (auto& ic : n->idx) {
        auto f = to.find(ic.dim);
        if (ic.dim_b >= 0 && to.count(ic.dim_b)) fail(":get through a Sparse map of an expression whose index mixes two iteration variables is not supported");
        if (f == to.end() || ic.sparse >= 0) continue;
        if (ic.dim_b >= 0 || ic.sign != 1) fail(":get through a Sparse map of an expression whose index mixes two iteration variables is not supported");
        if (f->second.sparse < 0) { ic.dim = f->second.dim; continue; }      // a rename: another variable over the same dimension (the component's own offset stays)
        if (ic.off != 0) fail(":get through a Sparse map of an expression that reads a shifted neighbour is not supported");
        ic = f->second;
    }
    for (auto& c : n->a) c = subst_expr(c, to, memo);
    E r = n; memo[e.get()] = r; return r;
}
void collect_dims(const E& e, std::vector<int>& dims, std::map<const Expr*, int>& seen)
{
    if (seen.count(e.get())) return; seen[e.get()] = 1;
    for (auto& ic : e->idx) for (int dd : { ic.dim, ic.dim2, ic.dim_b }) { if (dd < 0) continue; bool have = false; for (int d : dims) have = have || d == dd; if (!have) dims.push_back(dd); }
    for (auto& c : e->a) collect_dims(c, dims, seen);
}

// Sum({k, ...}, e) expanded (lib.t:146, thallo.t:5887-5922): the iteration variable over dimension d takes the constant value to[d] -- a component that loses its
// only variable becomes a constant index
E subst_const(const E& e, const std::map<int, int>& to, std::map<const Expr*, E>& memo)
{
    auto it = memo.find(e.get()); if (it != memo.end()) return it->second;
    auto n = std::make_shared<Expr>(*e);
    for (auto& ic : n->idx) {
        auto fa = to.find(ic.dim), fb = to.find(ic.dim_b), f2 = to.find(ic.dim2);
        if (ic.sparse >= 0) { if ((ic.dim >= 0 && fa != to.end()) || (ic.dim2 >= 0 && f2 != to.end())) fail("Sum over the index of a Sparse map is not supported"); continue; }
        if (ic.dim_b >= 0 && fb != to.end()) { add_off(ic.off, (long long)ic.sign_b * fb->second); ic.dim_b = -1; ic.sign_b = 1; }
        if (ic.dim >= 0 && fa != to.end()) { add_off(ic.off, (long long)ic.sign * fa->second); ic.dim = ic.dim_b; ic.sign = ic.dim_b >= 0 ? ic.sign_b : 1; ic.dim_b = -1; ic.sign_b = 1; }
    }
    for (auto& c : n->a) c = subst_const(c, to, memo);
    E r = n; memo[e.get()] = r; return r;
}

// ================================================================================================ interpreter
struct EnvRegistry {        // (a member of Interp in front of `globals`: set up before the first Env exists, torn down after the interpreter's own references went)
    std::set<Env*> live;
    EnvRegistry() { g_live_envs = &live; }
    void open_cycles()
    {
        std::vector<std::shared_ptr<Env>> hold;
        for (Env* e : live) if (auto sp = e->weak_from_this().lock()) hold.push_back(sp);
        for (auto& e : hold) { e->vars.clear(); e->parent.reset(); }
    }
    ~EnvRegistry() { open_cycles(); g_live_envs = nullptr; }
};
struct Interp {
    Problem& P;
    EnvRegistry env_registry;
    std::shared_ptr<Env> globals = std::make_shared<Env>();
    ~Interp() { globals.reset(); }
    int depth = 0;
    std::map<int, int> dim_calls;                  // how often each dimension was called for an iteration variable
    std::set<const Expr*> no_materialize;          // expressions the file asked to read inlined (e:set_materialize(false))
    std::vector<std::vector<const Expr*>> computed_keys;      // P.computed[k] was made from these component expressions
    struct ReturnEx { Values v; };
    struct BreakEx {};
    explicit Interp(Problem& p) : P(p) { install_builtins(); }

    // ---- symbolic value helpers
    static Value scalar(const E& e) { SymV s; s.k = SymV::Scalar; s.e = e; return Value::make_sym(s); }
    static Value vec(const std::vector<E>& v) { if (v.size() == 1) return scalar(v[0]); SymV s; s.k = SymV::Vec; s.v = v; return Value::make_sym(s); }
    static bool is_symk(const Value& v, SymV::K k) { return v.t == Value::Sym && v.sym->k == k; }
    // number / Scalar / Vec -> component list
    static bool as_comps(const Value& v, std::vector<E>& out)
    {
        if (v.t == Value::Num) { out = { konst(v.n) }; return true; }
        if (v.t == Value::Bool) { out = { konst(v.b ? 1.0 : 0.0) }; return true; }
        if (is_symk(v, SymV::Scalar)) { out = { v.sym->e }; return true; }
        if (is_symk(v, SymV::Vec)) { out = v.sym->v; return true; }
        if (is_symk(v, SymV::IndexDomain) || is_symk(v, SymV::IndexE)) return false;
        if (v.t == Value::Table) {                  // a Lua list of scalars, e.g. Residuals { reg = { e1, e2 } }
            out.clear();
            for (auto& x : v.tab->arr) { std::vector<E> c; if (!as_comps(x, c)) return false; out.insert(out.end(), c.begin(), c.end()); }
            return !out.empty();
        }
        return false;
    }
    static std::vector<E> comps(const Value& v, const char* what)
    { std::vector<E> c; if (!as_comps(v, c)) fail(std::string(what) + ": expected a number or an expression"); return c; }
    static E one(const Value& v, const char* what) { auto c = comps(v, what); if (c.size() != 1) fail(std::string(what) + ": expected a scalar"); return c[0]; }
    static IndexComp as_index(const Value& v, const char* what)
    {
        if (is_symk(v, SymV::IndexDomain)) { IndexComp ic; ic.dim = v.sym->id; return ic; }
        if (is_symk(v, SymV::IndexE)) return v.sym->ic;
        fail(std::string(what) + ": expected an index expression (an iteration variable plus an integer offset)");
    }
    static Value broadcast(Op op, const Value& a, const Value& b, const char* what)
    {
        std::vector<E> x = comps(a, what), y = comps(b, what), r;
        if (x.size() != y.size() && x.size() != 1 && y.size() != 1) fail(std::string(what) + ": vector lengths differ");
        const size_t n = std::max(x.size(), y.size());
        for (size_t i = 0; i < n; ++i) r.push_back(bin(op, x[x.size() == 1 ? 0 : i], y[y.size() == 1 ? 0 : i]));
        return vec(r);
    }
    static Value map1(Op op, const Value& a, const char* what) { std::vector<E> r; for (auto& e : comps(a, what)) r.push_back(un(op, e)); return vec(r); }

    Value arith(const std::string& op, const Value& a, const Value& b, int line)
    {
        if (a.t == Value::Num && b.t == Value::Num) {
            if (op == "+") return Value::num(a.n + b.n); if (op == "-") return Value::num(a.n - b.n); if (op == "*") return Value::num(a.n * b.n);
            if (op == "/") return Value::num(a.n / b.n); if (op == "%") return Value::num(a.n - std::floor(a.n / b.n) * b.n); if (op == "^") return Value::num(std::pow(a.n, b.n));
        }
        // index arithmetic: x + 1, x - dx, dx + x
        const bool ai = is_symk(a, SymV::IndexDomain) || is_symk(a, SymV::IndexE), bi = is_symk(b, SymV::IndexDomain) || is_symk(b, SymV::IndexE);
        if ((ai && b.t == Value::Num) || (bi && a.t == Value::Num && op == "+")) {
            IndexComp ic = as_index(ai ? a : b, "index arithmetic"); const double d = ai ? b.n : a.n;
            if (d != std::floor(d)) fail("line " + std::to_string(line) + ": non-integer index offset");
            if (ic.sparse >= 0 && d != 0) fail("line " + std::to_string(line) + ": offset on an index that went through a Sparse map");
            if (op == "+") add_off(ic.off, small_int(d, "index offset")); else if (op == "-") add_off(ic.off, -(long long)small_int(d, "index offset")); else fail("line " + std::to_string(line) + ": only + and - are defined on indices");
            SymV s; s.k = SymV::IndexE; s.ic = ic; return Value::make_sym(s);
        }
        if (ai && bi && (op == "+" || op == "-")) {   // two iteration variables: x - k + 8 (convolution.t, spatially_varying_deconvolution.t)
            const IndexComp x = as_index(a, "index arithmetic"), y = as_index(b, "index arithmetic");
            if (x.sparse >= 0 || y.sparse >= 0 || x.dim_b >= 0 || y.dim_b >= 0 || x.dim < 0 || y.dim < 0) fail("line " + std::to_string(line) + ": index arithmetic takes at most two iteration variables, none through a Sparse map");
            IndexComp ic = x;
            const int sg = op == "+" ? 1 : -1;
            ic.dim_b = y.dim; ic.sign_b = sg * y.sign; add_off(ic.off, (long long)sg * y.off);
            SymV r; r.k = SymV::IndexE; r.ic = ic; return Value::make_sym(r);
        }
        const Op o = op == "+" ? Op::Add : op == "-" ? Op::Sub : op == "*" ? Op::Mul : op == "/" ? Op::Div : op == "^" ? Op::Pow : Op::Const;
        if (o == Op::Const) fail("line " + std::to_string(line) + ": operator " + op + " on these operands");
        if (o == Op::Pow) {
            if (b.t != Value::Num || b.n != std::floor(b.n) || b.n < 1 || b.n > 8) fail("line " + std::to_string(line) + ": only small positive integer powers of expressions are supported");
            Value r = a; for (int i = 1; i < (int)b.n; ++i) r = broadcast(Op::Mul, r, a, "^"); return r;
        }
        return broadcast(o, a, b, ("line " + std::to_string(line) + " operator " + op).c_str());
    }

    // ---- evaluation
    Values eval_multi(const NP& n, const std::shared_ptr<Env>& env)
    {
        if (n->k == Node::Call || n->k == Node::Method) return call_node(n, env);
        return { eval(n, env) };
    }
    Values eval_list(const std::vector<NP>& es, const std::shared_ptr<Env>& env)
    {
        Values out;
        for (size_t i = 0; i < es.size(); ++i) {
            if (i + 1 == es.size()) { Values last = eval_multi(es[i], env); out.insert(out.end(), last.begin(), last.end()); }
            else out.push_back(eval(es[i], env));
        }
        return out;
    }
    Value eval(const NP& n, const std::shared_ptr<Env>& env)
    {
        switch (n->k) {
        case Node::Nil: return Value();
        case Node::True: return Value::boolean(true);
        case Node::False: return Value::boolean(false);
        case Node::Number: return Value::num(n->num);
        case Node::String: return Value::str(n->s);
        case Node::Paren: return eval(n->a[0], env);
        case Node::Function: { Value v; v.t = Value::Func; v.fn = std::make_shared<FuncV>(); v.fn->def = n; v.fn->env = env; return v; }
        case Node::Name: { Value* v = env->find(n->s); return v ? *v : Value(); }
        case Node::Index: return index_value(eval(n->a[0], env), eval(n->a[1], env), n->line);
        case Node::Call: case Node::Method: { Values r = call_node(n, env); return r.empty() ? Value() : r[0]; }
        case Node::Table: {
            Value v; v.t = Value::Table; v.tab = std::make_shared<TableV>();
            for (size_t i = 0; i < n->a.size(); ++i) {
                if (!n->keys[i]) {
                    if (i + 1 == n->a.size()) { Values last = eval_multi(n->a[i], env); for (auto& x : last) v.tab->arr.push_back(x); }
                    else v.tab->arr.push_back(eval(n->a[i], env));
                } else {
                    const Value key = eval(n->keys[i], env), val = eval(n->a[i], env);
                    table_set(v, key, val, n->line);
                }
            }
            return v;
        }
        case Node::Unop: {
            const Value x = eval(n->a[0], env);
            if (n->s == "not") return Value::boolean(!x.truthy());
            if (n->s == "#") { if (x.t == Value::Table) return Value::num((double)x.tab->arr.size()); if (x.t == Value::Str) return Value::num((double)x.s.size()); if (is_symk(x, SymV::Vec)) return Value::num((double)x.sym->v.size()); fail("line " + std::to_string(n->line) + ": # on this value"); }
            if (x.t == Value::Num) return Value::num(-x.n);
            return map1(Op::Neg, x, "unary minus");
        }
        case Node::Binop: {
            const std::string& op = n->s;
            if (op == "and") { const Value l = eval(n->a[0], env); return l.truthy() ? eval(n->a[1], env) : l; }
            if (op == "or") { const Value l = eval(n->a[0], env); return l.truthy() ? l : eval(n->a[1], env); }
            const Value l = eval(n->a[0], env), r = eval(n->a[1], env);
            if (op == "==" || op == "~=") {
                bool eq = false;
                if (l.t == r.t) { if (l.t == Value::Num) eq = l.n == r.n; else if (l.t == Value::Str) eq = l.s == r.s; else if (l.t == Value::Bool) eq = l.b == r.b; else if (l.t == Value::Nil) eq = true;
                                  else if (l.t == Value::Table) eq = l.tab == r.tab; else if (l.t == Value::Sym) eq = l.sym == r.sym; }
                return Value::boolean(op == "==" ? eq : !eq);
            }
            if (op == "<" || op == ">" || op == "<=" || op == ">=") {
                if (l.t != Value::Num || r.t != Value::Num) fail("line " + std::to_string(n->line) + ": " + op + " compares numbers only (use less / greater / ... on expressions)");
                return Value::boolean(op == "<" ? l.n < r.n : op == ">" ? l.n > r.n : op == "<=" ? l.n <= r.n : l.n >= r.n);
            }
            if (op == "..") { auto ts = [&](const Value& v) { if (v.t == Value::Str) return v.s; if (v.t == Value::Num) { char b[64]; snprintf(b, sizeof b, "%.14g", v.n); return std::string(b); } fail("concatenation of this value"); }; return Value::str(ts(l) + ts(r)); }
            return arith(op, l, r, n->line);
        }
        default: fail("line " + std::to_string(n->line) + ": not an expression");
        }
    }
    static Value* table_get(const Value& t, const Value& key)
    {
        if (key.t == Value::Num) { const double k = key.n; if (k == std::floor(k) && k >= 1 && k <= (double)t.tab->arr.size()) return &t.tab->arr[(size_t)k - 1]; return nullptr; }
        if (key.t == Value::Str) { for (auto& f : t.tab->fields) if (f.first == key.s) return &f.second; return nullptr; }
        return nullptr;
    }
    static void table_set(Value& t, const Value& key, const Value& val, int line)
    {
        if (key.t == Value::Num) {
            const double k = key.n;
            if (k != std::floor(k) || k < 1) fail("line " + std::to_string(line) + ": table key must be a positive integer or a string");
            if (k <= (double)t.tab->arr.size()) { t.tab->arr[(size_t)k - 1] = val; return; }
            if (k == (double)t.tab->arr.size() + 1) { t.tab->arr.push_back(val); return; }
            fail("line " + std::to_string(line) + ": sparse table");
        }
        if (key.t != Value::Str) fail("line " + std::to_string(line) + ": table key must be a positive integer or a string");
        for (auto& f : t.tab->fields) if (f.first == key.s) { f.second = val; return; }
        t.tab->fields.push_back({ key.s, val });
    }
    Value index_value(const Value& obj, const Value& key, int line)
    {
        if (obj.t == Value::Table) { Value* v = table_get(obj, key); return v ? *v : Value(); }
        if (is_symk(obj, SymV::Vec) || is_symk(obj, SymV::Scalar)) {           // normal[0]: 0-based component
            if (key.t != Value::Num) fail("line " + std::to_string(line) + ": vector index must be a number");
            auto c = comps(obj, "index"); const int i = small_int(key.n, "line " + std::to_string(line) + ": vector index");
            if (i < 0 || i >= (int)c.size()) fail("line " + std::to_string(line) + ": vector component " + std::to_string(i) + " out of range");
            return scalar(c[i]);
        }
        if (is_symk(obj, SymV::ResidualsH) && key.t == Value::Str) {
            for (size_t i = 0; i < P.residuals.size(); ++i) if (P.residuals[i].name == key.s) { SymV s; s.k = SymV::NamedRes; s.id = (int)i; return Value::make_sym(s); }
            fail("line " + std::to_string(line) + ": no residual named " + key.s);
        }
        if (is_symk(obj, SymV::NamedRes) && key.t == Value::Str) {
            if (key.s == "J" || key.s == "JtJ" || key.s == "Jp" || key.s == "JtF") { SymV s; s.k = SymV::MatInfo; s.id = obj.sym->id; s.s = key.s; return Value::make_sym(s); }
            fail("line " + std::to_string(line) + ": residual field " + key.s);
        }
        if (obj.t == Value::Nil) fail("line " + std::to_string(line) + ": attempt to index a nil value");
        fail("line " + std::to_string(line) + ": cannot index this value");
    }

    Values call_value(const Value& f, const Values& args, int line)
    {
        if (f.t == Value::Func) {
            if (f.fn->builtin >= 0) return call_builtin(f.fn->bname, args, line);
            if (++depth > 200) fail("recursion too deep");
            auto env = std::make_shared<Env>(); env->parent = f.fn->env;
            for (size_t i = 0; i < f.fn->def->names.size(); ++i) env->vars[f.fn->def->names[i]] = i < args.size() ? args[i] : Value();
            Values ret;
            try { exec_block(f.fn->def->a[0], env); } catch (ReturnEx& r) { ret = r.v; }
            --depth; return ret;
        }
        if (f.t == Value::Sym) return { call_sym(f, args, line) };
        fail("line " + std::to_string(line) + ": attempt to call a " + std::string(f.t == Value::Nil ? "nil" : "non-function") + " value");
    }
    Values call_node(const NP& n, const std::shared_ptr<Env>& env)
    {
        if (n->k == Node::Call) {
            const Value f = eval(n->a[0], env);
            std::vector<NP> argn(n->a.begin() + 1, n->a.end());
            if (f.t == Value::Nil) { std::string nm = n->a[0]->k == Node::Name ? n->a[0]->s : "?"; fail("line " + std::to_string(n->line) + ": '" + nm + "' is not defined (not part of the supported DSL subset)"); }
            return call_value(f, eval_list(argn, env), n->line);
        }
        const Value obj = eval(n->a[0], env);
        std::vector<NP> argn(n->a.begin() + 1, n->a.end());
        return call_method(obj, n->s, eval_list(argn, env), n->line);
    }

    // image(ix, iy) / sparse(e) / dim() / vec(i)
    Value call_sym(const Value& f, const Values& args, int line)
    {
        const std::string ln = "line " + std::to_string(line) + ": ";
        const SymV& s = *f.sym;
        if (s.k == SymV::Dim) {
            if (!args.empty()) fail(ln + "a dimension is called without arguments");
            // every call makes a new iteration variable (thallo.t:467-477).  The first one IS the dimension; further ones (k_0 = Kd(); k_1 = Kd()) are alias ids behind it
            P.dim_alias.resize(P.dims.size(), -1); P.dim_sizes.resize(P.dims.size(), -1);
            int& calls = dim_calls[s.id];
            SymV r; r.k = SymV::IndexDomain; r.id = s.id;
            if (calls++ > 0) {
                r.id = (int)P.dims.size();
                P.dims.push_back(P.dims[s.id] + "_" + std::to_string(calls)); P.dim_alias.push_back(s.id); P.dim_sizes.push_back(P.dim_sizes[s.id]);
            }
            return Value::make_sym(r);
        }
        if (s.k == SymV::Vec || s.k == SymV::Scalar) {
            if (args.size() != 1 || args[0].t != Value::Num) fail(ln + "vector component selection takes one number");
            return index_value(f, args[0], line);
        }
        if (s.k == SymV::Image) {
            const Input& in = P.inputs[s.id];
            if (in.kind == InputKind::Sparse) {
                const size_t nsrc = in.dims.size() - 1;                  // one or two source dimensions (thallo.t:1700-1740: Sparse(from, to, idx))
                if (args.size() != nsrc) fail(ln + in.name + " takes " + std::to_string(nsrc) + (nsrc == 1 ? " index" : " indices"));
                IndexComp ic;
                for (size_t k = 0; k < nsrc; ++k) {
                    const IndexComp a = as_index(args[k], in.name.c_str());
                    if (a.sparse >= 0) fail(ln + "nested Sparse maps are not supported");
                    // A map declared over one dimension and indexed with a variable of another (examples/bundle_fusion_solve: t_source = Sparse({CorrDim}, {T}) read as
                    // t_source(p), p = PairDim()): the reference does not compare the two (thallo.t:1977-1990 builds the access from whatever index it is given; the declared
                    // source space only sizes nothing on the device), the map is read at the variable's index.  Same here, with a note: the caller's array must hold an
                    // entry for every index of the variable's dimension.
                    if (P.canonical(a.dim) != in.dims[k])
                        fprintf(stderr, "[thallo] note: %s%s is declared over %s and indexed over %s: read at that variable's index, as in the reference\n", ln.c_str(), in.name.c_str(),
                                P.dims[in.dims[k]].c_str(), P.dims[P.canonical(a.dim)].c_str());
                    if (a.off != 0 || a.dim_b >= 0 || a.sign != 1) fail(ln + "offset inside a Sparse map access");
                    if (k == 0) ic.dim = a.dim; else ic.dim2 = a.dim;
                }
                ic.sparse = s.id;
                SymV r; r.k = SymV::IndexE; r.ic = ic; return Value::make_sym(r);
            }
            // image(ix, iy) or image(ix, iy, channel) (thallo.t:2000-2040: a trailing number selects the channel)
            int only_channel = -1;
            Values iargs = args;
            if (iargs.size() == in.dims.size() + 1 && iargs.back().t == Value::Num) {
                only_channel = (int)iargs.back().n; iargs.pop_back();
                if (only_channel < 0 || only_channel >= in.channels) fail(ln + in.name + ": channel out of range");
            }
            if (iargs.size() != in.dims.size()) fail(ln + in.name + " takes " + std::to_string(in.dims.size()) + " indices");
            std::vector<IndexComp> idx;
            for (size_t d = 0; d < iargs.size(); ++d) {
                IndexComp ic;
                if (iargs[d].t == Value::Num) { if (iargs[d].n != std::floor(iargs[d].n)) fail(ln + "non-integer constant index"); ic.off = small_int(iargs[d].n, ln + "constant index"); idx.push_back(ic); continue; }   // ConstantIndexComponent (thallo.t:485)
                ic = as_index(iargs[d], in.name.c_str());
                const int target = ic.sparse >= 0 ? P.inputs[ic.sparse].dims.back() : P.canonical(ic.dim);
                if (target != in.dims[d]) fail(ln + "index " + std::to_string(d) + " of " + in.name + " ranges over dimension " + P.dims[in.dims[d]] + ", got " + P.dims[target]);
                idx.push_back(ic);
            }
            std::vector<E> out;
            for (int c = 0; c < in.channels; ++c) { if (only_channel >= 0 && c != only_channel) continue; auto e = std::make_shared<Expr>(); e->op = Op::Load; e->input = s.id; e->channel = c; e->idx = idx; out.push_back(e); }
            return vec(out);
        }
        if (s.k == SymV::Sampled && s.s == "array") {                // A.SampledImageArray:__call(x, y, z, c), thallo.t:5887-5898.  Its partials are 0 in the reference
            const Input& in = P.inputs[s.id];                        // (op:getpartials returns { 0.0, 0.0 }, thallo.t:5913-5916): value only, like Constant(...)
            if (args.size() != 3 && args.size() != 4) fail(ln + "a sampled image array takes (x, y, z) or (x, y, z, channel)");
            const E x = un(Op::Detach, one(args[0], "sampled image array x")), y = un(Op::Detach, one(args[1], "sampled image array y")), z = un(Op::Detach, one(args[2], "sampled image array z"));
            int c0 = 0, c1 = in.channels;
            if (args.size() == 4) { if (args[3].t != Value::Num || args[3].n < 0 || args[3].n >= in.channels) fail(ln + "index out of bounds"); c0 = (int)args[3].n; c1 = c0 + 1; }
            std::vector<E> out;
            for (int c = c0; c < c1; ++c) { auto e = std::make_shared<Expr>(); e->op = Op::Sample; e->input = s.id; e->channel = c; e->a = { x, y, z }; out.push_back(e); }
            return out.size() == 1 ? scalar(out[0]) : vec(out);
        }
        if (s.k == SymV::Sampled) {                                  // A.SampledImage:__call(x, y, c), thallo.t:5784-5795
            const Input& in = P.inputs[s.id];
            if (args.size() != 2 && args.size() != 3) fail(ln + "a sampled image takes (x, y) or (x, y, channel)");
            const E x = one(args[0], "sampled image x"), y = one(args[1], "sampled image y");
            int c0 = 0, c1 = in.channels;
            if (args.size() == 3) { if (args[2].t != Value::Num || args[2].n < 0 || args[2].n >= in.channels) fail(ln + "index out of bounds"); c0 = (int)args[2].n; c1 = c0 + 1; }
            std::vector<E> out;
            for (int c = c0; c < c1; ++c) { auto e = std::make_shared<Expr>(); e->op = Op::Sample; e->input = s.id; e->input_dx = s.id_dx; e->input_dy = s.id_dy; e->channel = c; e->a = { x, y }; out.push_back(e); }
            return out.size() == 1 ? scalar(out[0]) : vec(out);
        }
        fail(ln + "this value is not callable");
    }

    Values call_method(const Value& obj, const std::string& m, const Values& args, int line)
    {
        const std::string ln = "line " + std::to_string(line) + ": ";
        if (is_symk(obj, SymV::IndexDomain) || is_symk(obj, SymV::IndexE)) {
            if (m == "asvalue") { auto e = std::make_shared<Expr>(); e->op = Op::IndexVal; e->idx = { as_index(obj, "asvalue") }; return { scalar(e) }; }      // (through a Sparse map: the map's entry as a value, thallo.t:1578 SparseAccess:asvalue)
        }
        if (is_symk(obj, SymV::Image)) {
            Input& in = P.inputs[obj.sym->id];
            if (m == "Exclude") { if (in.kind != InputKind::Unknown) fail(ln + "Exclude applies to unknowns"); in.exclude = one(args.at(0), "Exclude"); return {}; }
            if (m == "set_coherent") return {};
        }
        if (is_symk(obj, SymV::Vec) || is_symk(obj, SymV::Scalar)) {
            if (m == "slice") {                      // lib.t Slice on a value: components [a, b)
                if (args.size() != 2 || args[0].t != Value::Num || args[1].t != Value::Num) fail(ln + "slice(a, b)");
                auto c = comps(obj, "slice"); const int a = (int)args[0].n, b = (int)args[1].n;
                if (a < 0 || b > (int)c.size() || a >= b) fail(ln + "slice out of range");
                return { vec(std::vector<E>(c.begin() + a, c.begin() + b)) };
            }
            if (m == "get") {                        // computed-array access at shifted indices; 0 outside the iteration domain (thallo.t:876-883)
                auto c = comps(obj, "get");
                std::vector<int> dims; { std::map<const Expr*, int> seen; for (auto& e : c) collect_dims(e, dims, seen); }
                std::map<int, int> sh; std::vector<IndexComp> at;
                {   // POSITIONAL form (round 4): as many arguments as the expression has iteration variables (in the order the variables were made: t0, t1 = T(), T()) -- the
                    // k-th variable becomes the k-th argument, which may be a Sparse-mapped index, ANOTHER variable over the same dimension (a rename: f(t0):get(t1)), or
                    // the same variable with an offset (a shift, handled below).  examples/bundle_fusion_solve: M(t0, t1):get(t_target(p), t_source(p)) -- keyed by the map's
                    // target dimension, as the one-argument graph form below is, both variables would have received the last map.
                    std::vector<int> free = dims; std::sort(free.begin(), free.end());
                    bool positional = args.size() == free.size() && !free.empty(), pure_shift = true;
                    std::vector<IndexComp> ics;
                    for (size_t k = 0; positional && k < args.size(); ++k) {
                        IndexComp ic = as_index(args[k], "get"); ics.push_back(ic);
                        if (ic.dim_b >= 0 || ic.sign != 1) positional = false;
                        else if (ic.sparse >= 0) { pure_shift = false; if (ic.off != 0) fail(ln + ":get with an offset through a Sparse map"); if (P.inputs[ic.sparse].dims.back() != P.canonical(free[k])) positional = false; }
                        else if (ic.dim != free[k]) { pure_shift = false; if (P.canonical(ic.dim) != P.canonical(free[k])) positional = false; }
                    }
                    if (positional && !pure_shift) {
                        std::map<int, IndexComp> to; bool shifted = false;
                        for (size_t k = 0; k < ics.size(); ++k) { if (ics[k].sparse < 0 && ics[k].dim == free[k] && ics[k].off == 0) continue; to[free[k]] = ics[k]; shifted = shifted || (ics[k].sparse < 0 && ics[k].off != 0); }
                        if (shifted) fail(ln + ":get that renames a variable AND shifts it is not supported");
                        std::vector<E> out; std::map<const Expr*, E> memo;
                        for (auto& e : c) out.push_back(subst_expr(e, to, memo));
                        return { vec(out) };
                    }
                }
                {   // graph access: every argument went through a Sparse map -> substitute the mapped index for the map's target dimension
                    std::map<int, IndexComp> to; size_t nsp = 0;
                    for (auto& a : args) { IndexComp ic = as_index(a, "get"); if (ic.sparse >= 0) { ++nsp; if (ic.off != 0) fail(ln + ":get with an offset through a Sparse map"); to[P.inputs[ic.sparse].dims.back()] = ic; } }
                    if (nsp) {
                        if (nsp != args.size()) fail(ln + ":get mixes plain and Sparse-mapped indices");
                        std::vector<E> out; std::map<const Expr*, E> memo;
                        for (auto& e : c) out.push_back(subst_expr(e, to, memo));
                        return { vec(out) };
                    }
                }
                for (auto& a : args) { IndexComp ic = as_index(a, "get"); sh[ic.dim] = ic.off; at.push_back(ic); }
                bool any = false; for (auto& kv : sh) any = any || kv.second != 0;
                std::vector<E> out; std::map<const Expr*, E> memo;
                auto inb = std::make_shared<Expr>(); inb->op = Op::InBounds; inb->idx = at;
                for (auto& e : c) out.push_back(any ? mk(Op::Select, { inb, shift_expr(e, sh, memo), konst(0.0) }) : e);
                // Round 6: a pure shift of an expression over exactly the dimensions it is read at IS a computed-array access (thallo.t:1868-1937): the node keeps the inlined
                // form as its operand and names the array; the code generator materializes the array (value + gradient planes, a precompute kernel per GN iteration) unless
                // the file said set_materialize(false) / set_gradient_materialize(false) on the expression, or THALLO_FRONTEND_COMPUTED=0
                bool plain = at.size() == dims.size() && !dims.empty();
                for (size_t k = 0; plain && k < at.size(); ++k) {
                    plain = plain_index(at[k]) && std::find(dims.begin(), dims.end(), at[k].dim) != dims.end();
                    for (size_t j = 0; plain && j < k; ++j) plain = at[j].dim != at[k].dim;
                }
                for (auto& e : c) plain = plain && !no_materialize.count(e.get());
                if (!plain) { if (!any) return { obj }; return { vec(out) }; }
                std::vector<const Expr*> key; for (auto& e : c) key.push_back(e.get());
                int id = -1;
                for (size_t q = 0; q < computed_keys.size(); ++q) if (computed_keys[q] == key) id = (int)q;
                if (id < 0) {
                    id = (int)P.computed.size(); computed_keys.push_back(key);
                    ComputedArray ca; ca.exprs = c; for (auto& ic : at) ca.domain.push_back(ic.dim);
                    P.computed.push_back(ca);
                } else {
                    for (size_t k = 0; k < at.size(); ++k) if (P.computed[(size_t)id].domain[k] != at[k].dim) { if (!any) return { obj }; return { vec(out) }; }      // (read with its indices in another order: inlined)
                }
                std::vector<E> nodes;
                for (size_t k = 0; k < c.size(); ++k) {
                    auto n = std::make_shared<Expr>(); n->op = Op::Computed; n->input = id; n->channel = (int)k; n->idx = at; n->a = { out[k] };
                    nodes.push_back(n);
                }
                return { vec(nodes) };
            }
            if (m == "materialize" || m == "set_materialize" || m == "set_gradient_materialize") {      // computed-array scheduling hints (thallo.t:1907-1927): (false) = read it inlined
                const bool on = m == "materialize" || args.empty() || args[0].truthy();
                for (auto& e : comps(obj, "set_materialize")) { if (on) no_materialize.erase(e.get()); else no_materialize.insert(e.get()); }
                return {};
            }
            if (m == "dot") { if (args.size() != 1) fail(ln + "v:dot(w)"); return { scalar(dot(comps(obj, "dot"), comps(args[0], "dot"), ln)) }; }
        }
        if (is_symk(obj, SymV::MatInfo)) {
            Residual& r = P.residuals[obj.sym->id];
            if (m == "set_materialize") { const bool b = !args.empty() && args[0].truthy(); if (obj.sym->s == "J") r.mat_J = b; else if (obj.sym->s == "JtJ") r.mat_JtJ = b; else if (obj.sym->s == "Jp") r.mat_Jp = b; return {}; }
            if (m == "set_sparse" || m == "compute_at_output") return {};
        }
        if (is_symk(obj, SymV::NamedRes)) {
            // compute_at_output (thallo.t:5661-5674): map the residual at its OUTPUT, i.e. unknown-wise gather instead of residual-wise scatter; honoured where
            // the gather lowering exists (dsl_codegen.cpp), otherwise the plan says so in a warning.  reorder only permutes loop nests in the reference's
            // generated code (thallo.t:5690-5740): no effect on results, nothing to permute in a one-thread-per-element kernel.
            if (m == "compute_at_output") { P.residuals[obj.sym->id].at_output = args.empty() || args[0].truthy() ? 1 : 0; return { obj }; }
            if (m == "reorder" || m == "clear_reorder") return { obj };
        }
        if (is_symk(obj, SymV::ResidualsH)) {
            if (m == "set_direct_solve") { P.direct_solve = !args.empty() && args[0].truthy(); return {}; }      // thallo.t:5634-5636
            if (m == "merge") {                      // thallo.t:5676-5688: the second residual's expressions join the first
                if (args.size() != 2 || !is_symk(args[0], SymV::NamedRes) || !is_symk(args[1], SymV::NamedRes)) fail(ln + "merge(r1, r2)");
                const int i1 = args[0].sym->id, i2 = args[1].sym->id;
                if (i1 == i2) fail(ln + "merge of a residual with itself");
                if (P.residuals[i1].domain != P.residuals[i2].domain) fail(ln + "merge needs identical iteration domains");
                P.residuals[i1].name += "_" + P.residuals[i2].name;
                P.residuals[i1].exprs.insert(P.residuals[i1].exprs.end(), P.residuals[i2].exprs.begin(), P.residuals[i2].exprs.end());
                P.residuals[i2].exprs.clear();     // (kept as an empty slot so that handles stay valid; dropped at the end)
                return { args[0] };
            }
        }
        if (obj.t == Value::Table) {                 // t:insert(v)
            if (m == "insert" && args.size() == 1) { obj.tab->arr.push_back(args[0]); return {}; }
            Value* f = table_get(obj, Value::str(m));
            if (f) { Values a2{ obj }; a2.insert(a2.end(), args.begin(), args.end()); return call_value(*f, a2, line); }
        }
        fail(ln + "method '" + m + "' is not part of the supported DSL subset");
    }

    // ---- statements
    void exec_block(const NP& b, const std::shared_ptr<Env>& env) { for (auto& s : b->a) exec(s, env); }
    void assign(const NP& target, const Value& v, const std::shared_ptr<Env>& env)
    {
        if (target->k == Node::Name) { Value* slot = env->find(target->s); if (slot) *slot = v; else globals->vars[target->s] = v; return; }
        Value obj = eval(target->a[0], env); const Value key = eval(target->a[1], env);
        if (obj.t != Value::Table) fail("line " + std::to_string(target->line) + ": assignment into a non-table value");
        table_set(obj, key, v, target->line);
    }
    long steps = 0;     // statements executed: an energy file is a specification, not a program -- a budget turns `for i = 1, 1e12 do ... end` (or nested loops
                        // that each stay under their own guard) into an error instead of a hang inside Thallo_ProblemPlan
    void exec(const NP& s, const std::shared_ptr<Env>& env)
    {
        if (++steps > 20000000L) fail("the file executes more than 2e7 statements (an endless loop?)");
        switch (s->k) {
        case Node::Local: { Values v = eval_list(s->a, env); for (size_t i = 0; i < s->names.size(); ++i) env->vars[s->names[i]] = i < v.size() ? v[i] : Value(); return; }
        case Node::Assign: { Values v = eval_list(s->a, env); for (size_t i = 0; i < s->keys.size(); ++i) assign(s->keys[i], i < v.size() ? v[i] : Value(), env); return; }
        case Node::CallStat: call_node(s->a[0], env); return;
        case Node::Do: { auto e2 = std::make_shared<Env>(); e2->parent = env; exec_block(s->a[0], e2); return; }
        case Node::FuncStat: {
            Value f; f.t = Value::Func; f.fn = std::make_shared<FuncV>(); f.fn->def = s->a[1];
            if (s->is_local) { env->vars[s->a[0]->s] = Value(); f.fn->env = env; env->vars[s->a[0]->s] = f; }
            else { f.fn->env = env; assign(s->a[0], f, env); }
            return;
        }
        case Node::If: {
            for (size_t i = 0; i + 1 < s->a.size(); i += 2) {
                if (!s->a[i] || eval(s->a[i], env).truthy()) { auto e2 = std::make_shared<Env>(); e2->parent = env; exec_block(s->a[i + 1], e2); return; }
            }
            return;
        }
        case Node::While: {
            int guard = 0;
            try { while (eval(s->a[0], env).truthy()) { if (++guard > 1000000) fail("loop does not terminate"); auto e2 = std::make_shared<Env>(); e2->parent = env; exec_block(s->a[1], e2); } } catch (BreakEx&) {}
            return;
        }
        case Node::NumFor: {
            const Value a = eval(s->a[0], env), b = eval(s->a[1], env), c = s->a[2] ? eval(s->a[2], env) : Value::num(1);
            if (a.t != Value::Num || b.t != Value::Num || c.t != Value::Num || c.n == 0) fail("line " + std::to_string(s->line) + ": numeric for needs numbers");
            try { for (double i = a.n; c.n > 0 ? i <= b.n : i >= b.n; i += c.n) { auto e2 = std::make_shared<Env>(); e2->parent = env; e2->vars[s->names[0]] = Value::num(i); exec_block(s->a[3], e2); } } catch (BreakEx&) {}
            return;
        }
        case Node::GenFor: {
            std::vector<NP> es(s->a.begin(), s->a.end() - 1);
            Values it = eval_list(es, env);
            const NP body = s->a.back();
            auto run = [&](const Values& vals) { auto e2 = std::make_shared<Env>(); e2->parent = env; for (size_t i = 0; i < s->names.size(); ++i) e2->vars[s->names[i]] = i < vals.size() ? vals[i] : Value(); exec_block(body, e2); };
            try {
                if (!it.empty() && is_symk(it[0], SymV::StencilList)) {     // for dx, dy in Stencil { {1,0}, ... }
                    for (auto& row : it[0].sym->stencil) { Values v; for (double d : row) v.push_back(Value::num(d)); run(v); }
                } else if (it.size() >= 2 && it[0].t == Value::Func && (it[0].fn->bname == "ipairs_iter" || it[0].fn->bname == "pairs_iter") && it[1].t == Value::Table) {
                    const auto tab = it[1].tab;
                    for (size_t i = 0; i < tab->arr.size(); ++i) run({ Value::num((double)i + 1), tab->arr[i] });
                    if (it[0].fn->bname == "pairs_iter") for (auto& f : tab->fields) run({ Value::str(f.first), f.second });
                } else fail("line " + std::to_string(s->line) + ": for-in supports Stencil{...}, ipairs(t) and pairs(t)");
            } catch (BreakEx&) {}
            return;
        }
        case Node::Return: { ReturnEx r; r.v = eval_list(s->a, env); throw r; }
        case Node::Break: throw BreakEx();
        default: fail("line " + std::to_string(s->line) + ": statement not supported");
        }
    }

    // ---- builtins
    void def(const std::string& name) { Value f; f.t = Value::Func; f.fn = std::make_shared<FuncV>(); f.fn->builtin = 1; f.fn->bname = name; globals->vars[name] = f; }
    void install_builtins()
    {
        for (const char* n : { "Vec3", "RotationMatrixAndTranslationToMat4", "rigid_trans", "RodriguesSO3Exp", "PoseToMatrix", "Constant", "SampledImage", "SampledImageArray", "pow", "L_2_norm", "L_p", "gemv", "Dim", "neq", "Dims", "Inputs", "Unknown", "Array", "Image", "Sparse", "Param", "UsePreconditioner", "Residuals", "Stencil", "Select", "InBounds", "InBoundsExpanded", "Sum",
                               "eq", "greater", "greatereq", "less", "lesseq", "Not", "And", "Or", "All", "Any", "abs", "sqrt", "sin", "cos", "Vector", "dot", "cross",
                               "Rotate2D", "Rotate3D", "AngleAxisRotatePoint", "ipairs", "pairs", "print", "assert", "tostring", "tonumber", "unpack", "Sqrt", "normalize", "length",
                               "Mat4ToRigidTransform", "RigidTransformToMat4", "CameraToDepth", "Max", "Min", "matmul", "transpose", "InvertRigidTransform", "SelectOnAll" })
            def(n);
        for (const char* n : { "float", "float2", "float3", "float4", "float6", "float9", "thallo_float", "thallo_float2", "thallo_float3", "thallo_float4", "thallo_float6", "thallo_float9",
                               "thallo_mat3f", "thallo_mat4f", "mat3f", "mat4f", "float8", "thallo_float8", "uint8", "double", "int", "thallo_int" }) { SymV s; s.k = SymV::TypeName; s.s = n; globals->vars[n] = Value::make_sym(s); }
        Value math; math.t = Value::Table; math.tab = std::make_shared<TableV>();
        for (const char* n : { "sqrt", "sin", "cos", "abs" }) { Value f; f.t = Value::Func; f.fn = std::make_shared<FuncV>(); f.fn->builtin = 1; f.fn->bname = n; math.tab->fields.push_back({ n, f }); }
        math.tab->fields.push_back({ "pi", Value::num(3.14159265358979323846) });
        math.tab->fields.push_back({ "huge", Value::num(HUGE_VAL) });
        globals->vars["math"] = math;
        globals->vars["inf"] = Value::num(HUGE_VAL);                      // lib.t:16 (L.inf = math.huge)
        // `ad.*` as the energy files use it next to the lib.t names (ad.Vector, ad.sqrt, ad.select, ad.less ...): the same builtins
        Value adt; adt.t = Value::Table; adt.tab = std::make_shared<TableV>();
        const char* alias[][2] = { { "Vector", "Vector" }, { "sqrt", "sqrt" }, { "sin", "sin" }, { "cos", "cos" }, { "abs", "abs" }, { "pow", "pow" }, { "select", "Select" },
                                   { "less", "less" }, { "lesseq", "lesseq" }, { "greater", "greater" }, { "greatereq", "greatereq" }, { "eq", "eq" }, { "constant", "Constant" },
                                   { "and_", "And" }, { "or_", "Or" }, { "not_", "Not" } };
        for (auto& al : alias) { Value f; f.t = Value::Func; f.fn = std::make_shared<FuncV>(); f.fn->builtin = 1; f.fn->bname = al[1]; adt.tab->fields.push_back({ al[0], f }); }
        globals->vars["ad"] = adt;
    }
    static int type_channels(const Value& t, bool* u8, int line, bool* fixed32 = nullptr, bool* fixed64 = nullptr)
    {
        if (!is_symk(t, SymV::TypeName)) fail("line " + std::to_string(line) + ": expected an element type (float, thallo_float2, uint8, ...)");
        const std::string& s = t.sym->s; *u8 = s == "uint8";
        if (fixed32) *fixed32 = s.rfind("thallo_", 0) != 0 && s != "uint8" && s != "double" && s != "int";
        if (fixed64) *fixed64 = s == "double";
        if (s.find("mat3f") != std::string::npos) return 9;
        if (s.find("mat4f") != std::string::npos) return 16;
        const char last = s.back();
        return isdigit((unsigned char)last) && s != "uint8" ? last - '0' : 1;
    }
    std::vector<int> dim_list(const Value& t, int line)
    {
        if (t.t != Value::Table) fail("line " + std::to_string(line) + ": expected a list of dimensions {W, H}");
        std::vector<int> d; for (auto& v : t.tab->arr) { if (!is_symk(v, SymV::Dim)) fail("line " + std::to_string(line) + ": expected a dimension"); d.push_back(v.sym->id); }
        return d;
    }
    E cmp(Op op, const E& a, const E& b) { return mk(op, { a, b }); }
    Values call_builtin(const std::string& f, const Values& a, int line)
    {
        const std::string ln = "line " + std::to_string(line) + ": ";
        auto need = [&](size_t n) { if (a.size() < n) fail(ln + f + " needs " + std::to_string(n) + " argument(s)"); };
        if (f == "Dim") {                           // Dim("N", 0): one dimension with an explicit position in the dimensions array (thallo.t:1580-1600)
            need(2); if (a[0].t != Value::Str || a[1].t != Value::Num) fail(ln + "Dim(name, index)");
            if (!(a[1].n >= 0 && a[1].n <= 16 && a[1].n == std::floor(a[1].n))) fail(ln + "Dim index must be an integer in 0 .. 16"); const size_t id = (size_t)a[1].n;
            if (P.dims.size() <= id) P.dims.resize(id + 1);
            P.dims[id] = a[0].s; if (P.dim_sizes.size() <= id) P.dim_sizes.resize(id + 1, -1); if (P.dim_alias.size() <= id) P.dim_alias.resize(id + 1, -1); P.dim_sizes[id] = P.plan_dims ? (long)P.plan_dims[id] : -1;
            SymV s; s.k = SymV::Dim; s.id = (int)id; return { Value::make_sym(s) };
        }
        if (f == "Dims") { Values out; for (auto& v : a) { if (v.t != Value::Str) fail(ln + "Dims takes names"); SymV s; s.k = SymV::Dim; s.id = (int)P.dims.size(); P.dims.push_back(v.s); P.dim_alias.push_back(-1); P.dim_sizes.push_back(P.plan_dims ? (long)P.plan_dims[s.id] : -1); out.push_back(Value::make_sym(s)); } return out; }
        if (f == "Unknown" || f == "Array" || f == "Image") {           // Image: the deprecated spelling of Array (lib.t:573-576)
            need(3); Input in; in.kind = f == "Unknown" ? InputKind::Unknown : InputKind::Array;
            in.channels = type_channels(a[0], &in.is_u8, line, &in.fixed_f32, &in.fixed_f64); in.dims = dim_list(a[1], line);
            if (a[2].t != Value::Num) fail(ln + f + ": the third argument is the parameter index"); in.slot = small_int(a[2].n, ln + f + ": parameter index"); if (in.slot < 0 || in.slot > 4096) fail(ln + f + ": parameter index out of range");
            if (in.dims.empty() || in.dims.size() > 3) fail(ln + f + ": 1-, 2- and 3-dimensional images are supported");
            if (in.kind == InputKind::Unknown && in.is_u8) fail(ln + "uint8 unknowns are not supported");
            return { decl(in) };
        }
        if (f == "Sparse") { need(3); Input in; in.kind = InputKind::Sparse; auto from = dim_list(a[0], line), to = dim_list(a[1], line); if (from.empty() || from.size() > 2 || to.size() != 1) fail(ln + "Sparse({E}, {N}, idx) or Sparse({W,H}, {N}, idx)"); in.dims = from; in.dims.push_back(to[0]); if (a[2].t != Value::Num) fail(ln + "Sparse: the third argument is the parameter index"); in.slot = small_int(a[2].n, ln + "Sparse: parameter index"); if (in.slot < 0 || in.slot > 4096) fail(ln + "Sparse: parameter index out of range"); return { decl(in) }; }
        if (f == "Param") { need(2); Input in; in.kind = InputKind::Param; bool u8; if (type_channels(a[0], &u8, line, &in.fixed_f32, &in.fixed_f64) != 1 || u8) fail(ln + "scalar float Params are supported"); if (a[1].t != Value::Num) fail(ln + "Param(type, index)"); in.slot = small_int(a[1].n, ln + "Param: parameter index"); if (in.slot < 0 || in.slot > 4096) fail(ln + "Param: parameter index out of range"); return { decl(in) }; }
        if (f == "Inputs") {
            need(1); if (a[0].t != Value::Table) fail(ln + "Inputs { name = ..., ... }");
            for (auto& kv : a[0].tab->fields) {
                if (!is_symk(kv.second, SymV::Image)) fail(ln + "Inputs: " + kv.first + " is not an Unknown / Array / Sparse / Param");
                Input& in = P.inputs[kv.second.sym->id]; in.name = kv.first;
                if (in.kind == InputKind::Param) { auto e = std::make_shared<Expr>(); e->op = Op::Param; e->input = kv.second.sym->id; globals->vars[kv.first] = scalar(e); }
                else globals->vars[kv.first] = kv.second;
                P.max_slot = std::max(P.max_slot, in.slot);
            }
            return {};
        }
        if (f == "UsePreconditioner") { P.use_preconditioner = !a.empty() && a[0].truthy(); return {}; }
        if (f == "Stencil") { need(1); SymV s; s.k = SymV::StencilList; if (a[0].t != Value::Table) fail(ln + "Stencil { {dx, dy}, ... }"); for (auto& row : a[0].tab->arr) { std::vector<double> r; if (row.t != Value::Table) fail(ln + "Stencil entries are lists"); for (auto& v : row.tab->arr) { if (v.t != Value::Num) fail(ln + "Stencil offsets are numbers"); r.push_back(v.n); } s.stencil.push_back(r); } return { Value::make_sym(s) }; }
        if (f == "Residuals") {
            need(1); if (a[0].t != Value::Table) fail(ln + "Residuals { name = expr, ... }");
            for (auto& kv : a[0].tab->fields) {
                Residual r; r.name = kv.first; r.exprs = comps(kv.second, ("residual " + kv.first).c_str());
                std::map<const Expr*, int> seen; for (auto& e : r.exprs) collect_dims(e, r.domain, seen);
                if (r.domain.empty()) fail(ln + "residual " + kv.first + " does not depend on any iteration variable");
                P.residuals.push_back(r);
            }
            if (!a[0].tab->arr.empty()) fail(ln + "Residuals entries need names");
            SymV s; s.k = SymV::ResidualsH; return { Value::make_sym(s) };
        }
        if (f == "Select") {                        // ad.t:800-809
            need(3); auto c = comps(a[0], "Select"), x = comps(a[1], "Select"), y = comps(a[2], "Select");
            const size_t n = std::max(c.size(), std::max(x.size(), y.size())); std::vector<E> out;       // (a vector condition broadcasts scalar branches)
            if ((c.size() != 1 && c.size() != n) || (x.size() != n && x.size() != 1) || (y.size() != n && y.size() != 1)) fail(ln + "Select: vector lengths differ");
            for (size_t i = 0; i < n; ++i) out.push_back(mk(Op::Select, { c[c.size() == 1 ? 0 : i], x[x.size() == 1 ? 0 : i], y[y.size() == 1 ? 0 : i] }));
            return { vec(out) };
        }
        if (f == "Sum") {                           // lib.t:146 Sum({k, ...}, e) = P:TensorContraction: e summed over the listed iteration variables
            need(2);
            if (a[0].t != Value::Table || a[0].tab->arr.empty()) fail(ln + "Sum({k, ...}, expression)");
            if (!P.plan_dims) fail(ln + "Sum needs the sizes of the problem's dimensions: it is expanded when the problem is planned (Thallo_ProblemPlan)");
            std::vector<int> ds; long terms = 1;
            for (auto& v : a[0].tab->arr) {
                if (!is_symk(v, SymV::IndexDomain)) fail(ln + "Sum: the first argument lists iteration variables");
                const int d = v.sym->id;
                if (d < 0 || d >= (int)P.dim_sizes.size() || P.dim_sizes[d] < 0) fail(ln + "Sum over a dimension without a size");
                for (int e : ds) if (e == d) fail(ln + "Sum: a dimension listed twice");
                ds.push_back(d); terms *= P.dim_sizes[d];
            }
            if (terms < 1 || terms > 4096) fail(ln + "Sum over " + std::to_string(terms) + " terms (expanded at Plan time: at most 4096)");
            const std::vector<E> body = comps(a[1], "Sum");
            std::vector<E> acc(body.size());
            std::vector<int> at(ds.size(), 0);
            for (long t = 0; t < terms; ++t) {
                std::map<int, int> to; for (size_t k = 0; k < ds.size(); ++k) to[ds[k]] = at[k];
                std::map<const Expr*, E> memo;
                for (size_t c = 0; c < body.size(); ++c) { const E term = subst_const(body[c], to, memo); acc[c] = t == 0 ? term : bin(Op::Add, acc[c], term); }
                for (size_t k = 0; k < ds.size(); ++k) { if (++at[k] < P.dim_sizes[ds[k]]) break; at[k] = 0; }      // first listed variable fastest
            }
            return { vec(acc) };
        }
        if (f == "InBounds" || f == "InBoundsExpanded") {
            auto e = std::make_shared<Expr>(); e->op = Op::InBounds;
            size_t n = a.size();
            if (f == "InBoundsExpanded") { need(2); if (a.back().t != Value::Num) fail(ln + "InBoundsExpanded(x, y, n)"); e->expand = (int)a.back().n; --n; }
            for (size_t i = 0; i < n; ++i) e->idx.push_back(as_index(a[i], f.c_str()));
            if (e->idx.empty()) fail(ln + f + " needs indices");
            return { scalar(e) };
        }
        auto cmp2 = [&](Op op) { need(2); auto x = comps(a[0], f.c_str()), y = comps(a[1], f.c_str()); const size_t n = std::max(x.size(), y.size()); std::vector<E> out;
                                 if (x.size() != y.size() && x.size() != 1 && y.size() != 1) fail(ln + f + ": vector lengths differ");
                                 for (size_t i = 0; i < n; ++i) out.push_back(cmp(op, x[x.size() == 1 ? 0 : i], y[y.size() == 1 ? 0 : i])); return Values{ vec(out) }; };
        if (f == "neq") { Values e = cmp2(Op::Eq); return { map1(Op::Not, e[0], "neq") }; }
        if (f == "eq") return cmp2(Op::Eq); if (f == "greater") return cmp2(Op::Gt); if (f == "greatereq") return cmp2(Op::Ge);
        if (f == "less") return cmp2(Op::Lt); if (f == "lesseq") return cmp2(Op::Le);
        if (f == "And") return cmp2(Op::And); if (f == "Or") return cmp2(Op::Or);
        if (f == "Not") { need(1); return { map1(Op::Not, a[0], "Not") }; }
        if (f == "All" || f == "Any") { need(1); auto c = comps(a[0], f.c_str()); E r = c[0]; for (size_t i = 1; i < c.size(); ++i) r = mk(f == "All" ? Op::And : Op::Or, { r, c[i] }); return { scalar(r) }; }
        if (f == "SampledImage") {                                   // lib.t:144 = ad.sampledimage(image, imagedx, imagedy), thallo.t:5802-5818
            const int n = (int)a.size(); if (n != 1 && n != 3) fail(ln + "SampledImage(image) or SampledImage(image, dx, dy)");
            SymV r; r.k = SymV::Sampled;
            for (int k = 0; k < n; ++k) {
                if (!is_symk(a[k], SymV::Image)) fail(ln + "expected an image or a sampled image as a derivative");
                const Input& in = P.inputs[a[k].sym->id];
                if (in.kind != InputKind::Array || in.is_u8) fail(ln + "sampled images are float Arrays");
                if (in.dims.size() != 2) fail(ln + "sampled images must be 2D");
                if (k && in.channels != P.inputs[a[0].sym->id].channels) fail(ln + "the derivative images of a sampled image have its channel count");
                (k == 0 ? r.id : k == 1 ? r.id_dx : r.id_dy) = a[k].sym->id;
            }
            return { Value::make_sym(r) };
        }
        if (f == "SampledImageArray") {                             // lib.t:145 = ad.sampledimagearray(image), thallo.t:5887-5922: a 3-D Array sampled at (x, y) in layer z
            need(1);
            if (!is_symk(a[0], SymV::Image)) fail(ln + "SampledImageArray(image)");
            const Input& in = P.inputs[a[0].sym->id];
            if (in.kind != InputKind::Array || in.is_u8) fail(ln + "sampled image arrays are float Arrays");
            if (in.dims.size() != 3) fail(ln + "sampled image arrays must be 3D");
            SymV r; r.k = SymV::Sampled; r.id = a[0].sym->id; r.s = "array";
            return { Value::make_sym(r) };
        }
        if (f == "Constant") { need(1); return { map1(Op::Detach, a[0], "Constant") }; }                     // lib.t:194 (ad.constant)
        if (f == "pow") {                                                                                        // ad.pow(base, exponent): exponent a number, a Param or any derivative-free expression
            need(2); auto bs = comps(a[0], "pow"), ex = comps(a[1], "pow");
            if (ex.size() != 1) fail(ln + "pow: scalar exponent expected");
            std::vector<E> out; for (auto& b : bs) out.push_back(mk(Op::Pow, { b, ex[0] }));
            return { vec(out) };
        }
        if (f == "L_2_norm") { need(1); auto v = comps(a[0], "L_2_norm"); if (v.size() == 1) return { a[0] }; return { scalar(un(Op::Sqrt, dot(v, v, ln))) }; }      // lib.t:148-155
        if (f == "L_p") {                                                                                        // lib.t:157-169: sqrt((|v| + eps)^(p-2)) held constant, times v
            if (a.size() < 2) fail(ln + "L_p(val, p, domains)");
            auto v = comps(a[0], "L_p"), pe = comps(a[1], "L_p");
            if (pe.size() != 1) fail(ln + "L_p: scalar p expected");
            E dist = v.size() == 1 ? v[0] : un(Op::Sqrt, dot(v, v, ln));
            E cw = mk(Op::Detach, { un(Op::Sqrt, mk(Op::Pow, { bin(Op::Add, dist, konst(0.0000001)), bin(Op::Sub, pe[0], konst(2.0)) })) });
            std::vector<E> out; for (auto& c : v) out.push_back(bin(Op::Mul, cw, c));
            return { vec(out) };
        }
        if (f == "abs") { need(1); if (a[0].t == Value::Num) return { Value::num(std::fabs(a[0].n)) }; return { map1(Op::Abs, a[0], "abs") }; }
        if (f == "sqrt" || f == "Sqrt") { need(1); if (a[0].t == Value::Num) return { Value::num(std::sqrt(a[0].n)) }; return { map1(Op::Sqrt, a[0], "sqrt") }; }
        if (f == "sin") { need(1); if (a[0].t == Value::Num) return { Value::num(std::sin(a[0].n)) }; return { map1(Op::Sin, a[0], "sin") }; }
        if (f == "cos") { need(1); if (a[0].t == Value::Num) return { Value::num(std::cos(a[0].n)) }; return { map1(Op::Cos, a[0], "cos") }; }
        if (f == "Vector") { std::vector<E> out; for (auto& v : a) { auto c = comps(v, "Vector"); out.insert(out.end(), c.begin(), c.end()); } if (out.empty()) fail(ln + "empty Vector"); SymV s; s.k = SymV::Vec; s.v = out; return { Value::make_sym(s) }; }
        if (f == "gemv") {                          // lib.t:78-90: row-major matrix (rows x cols values) times a vector of cols values
            need(2); auto M = comps(a[0], "gemv"), v = comps(a[1], "gemv");
            if (v.empty() || M.size() % v.size()) fail(ln + "gemv: matrix size is not a multiple of the vector size");
            std::vector<E> out;
            for (size_t r = 0; r < M.size() / v.size(); ++r) { E val = bin(Op::Mul, M[r * v.size()], v[0]); for (size_t c = 1; c < v.size(); ++c) val = bin(Op::Add, val, bin(Op::Mul, M[r * v.size() + c], v[c])); out.push_back(val); }
            return { vec(out) };
        }
        if (f == "dot") { need(2); return { scalar(dot(comps(a[0], "dot"), comps(a[1], "dot"), ln)) }; }
        if (f == "cross") { need(2); return { vec(cross(comps(a[0], "cross"), comps(a[1], "cross"), ln)) }; }
        if (f == "normalize") { need(1); auto v = comps(a[0], "normalize"); E inv = bin(Op::Div, konst(1.0), un(Op::Sqrt, dot(v, v, ln))); std::vector<E> out; for (auto& e : v) out.push_back(bin(Op::Mul, e, inv)); return { vec(out) }; }
        if (f == "length") { need(2); auto x = comps(a[0], "length"), y = comps(a[1], "length"); if (x.size() != y.size()) fail(ln + "length: sizes differ"); std::vector<E> d; for (size_t i = 0; i < x.size(); ++i) d.push_back(bin(Op::Sub, x[i], y[i])); return { scalar(un(Op::Sqrt, dot(d, d, ln))) }; }
        if (f == "Rotate2D") {                      // lib.t:138-142
            need(2); E ang = one(a[0], "Rotate2D"); auto v = comps(a[1], "Rotate2D"); if (v.size() != 2) fail(ln + "Rotate2D rotates 2-vectors");
            E c = un(Op::Cos, ang), s = un(Op::Sin, ang);
            return { vec({ bin(Op::Add, bin(Op::Mul, c, v[0]), bin(Op::Mul, un(Op::Neg, s), v[1])), bin(Op::Add, bin(Op::Mul, s, v[0]), bin(Op::Mul, c, v[1])) }) };
        }
        if (f == "Rotate3D") {                      // lib.t:123-137 (ZYX Euler), then gemv
            need(2); auto an = comps(a[0], "Rotate3D"), v = comps(a[1], "Rotate3D"); if (an.size() != 3 || v.size() != 3) fail(ln + "Rotate3D(angle3, vector3)");
            E ca = un(Op::Cos, an[0]), cb = un(Op::Cos, an[1]), cg = un(Op::Cos, an[2]), sa = un(Op::Sin, an[0]), sb = un(Op::Sin, an[1]), sg = un(Op::Sin, an[2]);
            auto M = [&](E x, E y) { return bin(Op::Mul, x, y); }; auto A = [&](E x, E y) { return bin(Op::Add, x, y); }; auto N = [&](E x) { return un(Op::Neg, x); };
            std::vector<E> m = { M(cg, cb), A(M(N(sg), ca), M(M(cg, sb), sa)), A(M(sg, sa), M(M(cg, sb), ca)),
                                 M(sg, cb), A(M(cg, ca), M(M(sg, sb), sa)), A(M(N(cg), sa), M(M(sg, sb), ca)),
                                 N(sb), M(cb, sa), M(cb, ca) };
            std::vector<E> out; for (int r = 0; r < 3; ++r) out.push_back(A(A(M(m[3 * r], v[0]), M(m[3 * r + 1], v[1])), M(m[3 * r + 2], v[2])));
            return { vec(out) };
        }
        if (f == "Vec3") { need(1); auto v = comps(a[0], "Vec3"); if (v.size() < 3) fail(ln + "Vec3 of a shorter vector"); return { vec({ v[0], v[1], v[2] }) }; }
        if (f == "RotationMatrixAndTranslationToMat4") {      // lib.t:256-261: [R | t ; 0 0 0 1], row-major
            need(2); auto r = comps(a[0], f.c_str()), t = comps(a[1], f.c_str()); if (r.size() != 9 || t.size() != 3) fail(ln + f + "(matrix9, vector3)");
            return { vec({ r[0], r[1], r[2], t[0], r[3], r[4], r[5], t[1], r[6], r[7], r[8], t[2], konst(0.0), konst(0.0), konst(0.0), konst(1.0) }) };
        }
        if (f == "Mat4ToRigidTransform") {          // lib.t:263-267: the first three rows of a 4x4
            need(1); auto m = comps(a[0], f.c_str()); if (m.size() != 16) fail(ln + f + "(matrix16)");
            return { vec(std::vector<E>(m.begin(), m.begin() + 12)) };
        }
        if (f == "RigidTransformToMat4") {          // lib.t:269-274
            need(1); auto m = comps(a[0], f.c_str()); if (m.size() != 12) fail(ln + f + "(matrix12)");
            m.push_back(konst(0.0)); m.push_back(konst(0.0)); m.push_back(konst(0.0)); m.push_back(konst(1.0));
            return { vec(m) };
        }
        if (f == "CameraToDepth") {                 // lib.t:276-280: pinhole projection (x fx / z + cx, y fy / z + cy)
            need(5); auto pos = comps(a[4], f.c_str()); if (pos.size() < 3) fail(ln + "CameraToDepth(fx, fy, cx, cy, position3)");
            E fx = one(a[0], f.c_str()), fy = one(a[1], f.c_str()), cx = one(a[2], f.c_str()), cy = one(a[3], f.c_str());
            return { vec({ bin(Op::Add, bin(Op::Div, bin(Op::Mul, pos[0], fx), pos[2]), cx), bin(Op::Add, bin(Op::Div, bin(Op::Mul, pos[1], fy), pos[2]), cy) }) };
        }
        if (f == "Max" || f == "Min") {             // lib.t:282-285: select(greater(a, b), a, b)
            need(2);
            if (a[0].t == Value::Num && a[1].t == Value::Num) return { Value::num(f == "Max" ? std::fmax(a[0].n, a[1].n) : std::fmin(a[0].n, a[1].n)) };
            E x = one(a[0], f.c_str()), y = one(a[1], f.c_str());
            return { scalar(mk(Op::Select, { mk(f == "Max" ? Op::Gt : Op::Lt, { x, y }), x, y })) };
        }
        if (f == "matmul" || f == "transpose") {    // lib.t:287-305, 440-452: square row-major matrices
            auto A_ = comps(a[0], f.c_str());
            size_t dim = 0; while (dim * dim < A_.size()) ++dim;
            if (dim * dim != A_.size()) fail(ln + f + ": square matrices only");
            std::vector<E> out;
            if (f == "transpose") { need(1); for (size_t i = 0; i < dim; ++i) for (size_t j = 0; j < dim; ++j) out.push_back(A_[j * dim + i]); return { vec(out) }; }
            need(2); auto B_ = comps(a[1], f.c_str()); if (B_.size() != A_.size()) fail(ln + "matmul: sizes differ");
            for (size_t i = 0; i < dim; ++i) for (size_t j = 0; j < dim; ++j) {
                E c = konst(0.0);
                for (size_t k = 0; k < dim; ++k) c = bin(Op::Add, c, bin(Op::Mul, A_[i * dim + k], B_[k * dim + j]));
                out.push_back(c);
            }
            return { vec(out) };
        }
        if (f == "InvertRigidTransform") {          // lib.t:454-464: [R | t]^-1 = [R^T | -R^T t] of a 4x4
            need(1); auto m = comps(a[0], f.c_str()); if (m.size() != 16) fail(ln + f + "(matrix16)");
            const E Rt[9] = { m[0], m[4], m[8], m[1], m[5], m[9], m[2], m[6], m[10] }, t[3] = { m[3], m[7], m[11] };
            std::vector<E> out;
            for (int r = 0; r < 3; ++r) {
                E nt = bin(Op::Mul, un(Op::Neg, Rt[3 * r]), t[0]);
                for (int c = 1; c < 3; ++c) nt = bin(Op::Add, nt, bin(Op::Mul, un(Op::Neg, Rt[3 * r + c]), t[c]));
                out.push_back(Rt[3 * r]); out.push_back(Rt[3 * r + 1]); out.push_back(Rt[3 * r + 2]); out.push_back(nt);
            }
            out.push_back(konst(0.0)); out.push_back(konst(0.0)); out.push_back(konst(0.0)); out.push_back(konst(1.0));
            return { vec(out) };
        }
        if (f == "SelectOnAll") {                   // lib.t:196-205: val where every predicate of the list holds, else default
            need(3);
            if (a[0].t != Value::Table || a[0].tab->arr.empty()) fail(ln + "SelectOnAll({predicates}, value, default)");
            std::vector<E> preds; for (auto& pv : a[0].tab->arr) preds.push_back(one(pv, f.c_str()));
            if (preds.empty()) fail(ln + "SelectOnAll() requires at least one predicate");
            auto val = comps(a[1], f.c_str()), dflt = comps(a[2], f.c_str());
            if (dflt.size() == 1 && val.size() > 1) dflt.assign(val.size(), dflt[0]);
            if (dflt.size() != val.size()) fail(ln + "SelectOnAll: value and default differ in size");
            std::vector<E> out;
            for (size_t c = 0; c < val.size(); ++c) { E r = val[c]; for (size_t i = preds.size(); i-- > 0;) r = mk(Op::Select, { preds[i], r, dflt[c] }); out.push_back(r); }
            return { vec(out) };
        }
        if (f == "rigid_trans") {                   // lib.t:508-510: the first three rows of M (4x4, row-major) applied to (v, 1)
            need(2); auto M = comps(a[0], f.c_str()), v = comps(a[1], f.c_str()); if (((M.size() != 16) && (M.size() != 12)) || v.size() < 3) fail(ln + "rigid_trans(matrix16 or matrix12, vector3)");      // (gemv on the 3 x 4 rigid form gives the same three rows)
            std::vector<E> out;
            for (int r = 0; r < 3; ++r) out.push_back(bin(Op::Add, bin(Op::Add, bin(Op::Add, bin(Op::Mul, M[4 * r], v[0]), bin(Op::Mul, M[4 * r + 1], v[1])), bin(Op::Mul, M[4 * r + 2], v[2])), bin(Op::Mul, M[4 * r + 3], konst(1.0))));
            return { vec(out) };
        }
        if (f == "RodriguesSO3Exp" || f == "PoseToMatrix") {
            // lib.t:207-240 / 466-501: rotation vector w -> R = I + A [w]x + B [w]x^2 written out entry by entry; PoseToMatrix picks A, B (and the translation's
            // V-matrix coefficients) by the size of |w|^2 -- Taylor forms below 1e-8 and 1e-6, closed forms above -- and returns the 4x4 pose
            auto M = [&](E x, E y) { return bin(Op::Mul, x, y); }; auto A_ = [&](E x, E y) { return bin(Op::Add, x, y); }; auto S_ = [&](E x, E y) { return bin(Op::Sub, x, y); };
            auto rodrigues = [&](const std::vector<E>& w, E A, E B) {
                E wx2 = M(w[0], w[0]), wy2 = M(w[1], w[1]), wz2 = M(w[2], w[2]);
                E R00 = S_(konst(1.0), M(B, A_(wy2, wz2))), R11 = S_(konst(1.0), M(B, A_(wx2, wz2))), R22 = S_(konst(1.0), M(B, A_(wx2, wy2)));
                E a = M(A, w[2]), b = M(B, M(w[0], w[1])); E R01 = S_(b, a), R10 = A_(b, a);
                a = M(A, w[1]); b = M(B, M(w[0], w[2])); E R02 = A_(b, a), R20 = S_(b, a);
                a = M(A, w[0]); b = M(B, M(w[1], w[2])); E R12 = S_(b, a), R21 = A_(b, a);
                return std::vector<E>{ R00, R01, R02, R10, R11, R12, R20, R21, R22 };
            };
            if (f == "RodriguesSO3Exp") { need(3); auto w = comps(a[0], f.c_str()); if (w.size() != 3) fail(ln + "RodriguesSO3Exp(vector3, A, B)"); return { vec(rodrigues(w, one(a[1], f.c_str()), one(a[2], f.c_str()))) }; }
            need(2); auto rot = comps(a[0], f.c_str()), tr = comps(a[1], f.c_str()); if (rot.size() != 3 || tr.size() != 3) fail(ln + "PoseToMatrix(rotation3, translation3)");
            E th2 = dot(rot, rot, ln), th = un(Op::Sqrt, th2);
            auto cr = cross(rot, tr, ln), wcr = cross(rot, cr, ln);
            E small = mk(Op::Lt, { th2, konst(1e-8) }), mid = mk(Op::Lt, { th2, konst(1e-6) });
            const double sixth = 1.0 / 6.0, twentieth = 1.0 / 20.0;
            E A_s = S_(konst(1.0), M(konst(sixth), th2)), B_s = konst(0.5);
            E C_m = M(konst(sixth), S_(konst(1.0), M(konst(twentieth), th2))), A_m = S_(konst(1.0), M(th2, C_m)), B_m = S_(konst(0.5), M(konst(0.25 * sixth), th2));
            E inv = bin(Op::Div, konst(1.0), th), inv2 = M(inv, inv);
            E A_l = M(un(Op::Sin, th), inv), B_l = M(S_(konst(1.0), un(Op::Cos, th)), inv2), C_l = M(S_(konst(1.0), A_l), inv2);
            auto sel = [&](E c, E x, E y) { return mk(Op::Select, { c, x, y }); };
            std::vector<E> t3;
            for (int i = 0; i < 3; ++i) {
                E ts = A_(tr[i], M(konst(0.5), cr[i])), tm = A_(A_(tr[i], M(B_m, cr[i])), M(C_m, wcr[i])), tl = A_(A_(tr[i], M(B_l, cr[i])), M(C_l, wcr[i]));
                t3.push_back(sel(small, ts, sel(mid, tm, tl)));
            }
            E A = sel(small, A_s, sel(mid, A_m, A_l)), B = sel(small, B_s, sel(mid, B_m, B_l));
            auto R = rodrigues(rot, A, B);
            return { vec({ R[0], R[1], R[2], t3[0], R[3], R[4], R[5], t3[1], R[6], R[7], R[8], t3[2], konst(0.0), konst(0.0), konst(0.0), konst(1.0) }) };
        }
        if (f == "AngleAxisRotatePoint") {          // lib.t:514-555
            need(2); auto w0 = comps(a[0], f.c_str()), pt = comps(a[1], f.c_str()); if (w0.size() != 3 || pt.size() != 3) fail(ln + "AngleAxisRotatePoint(axis3, point3)");
            E th2 = dot(w0, w0, ln), large = cmp(Op::Gt, th2, konst(1e-8)), th = un(Op::Sqrt, th2), ct = un(Op::Cos, th), st = un(Op::Sin, th), inv = bin(Op::Div, konst(1.0), th);
            std::vector<E> w; for (auto& e : w0) w.push_back(bin(Op::Mul, e, inv));
            auto wxp = cross(w, pt, ln); E tmp = bin(Op::Mul, dot(w, pt, ln), bin(Op::Sub, konst(1.0), ct));
            auto sxp = cross(w0, pt, ln);
            std::vector<E> out;
            for (int i = 0; i < 3; ++i) {
                E lg = bin(Op::Add, bin(Op::Add, bin(Op::Mul, pt[i], ct), bin(Op::Mul, wxp[i], st)), bin(Op::Mul, w[i], tmp));
                out.push_back(mk(Op::Select, { large, lg, bin(Op::Add, pt[i], sxp[i]) }));
            }
            return { vec(out) };
        }
        if (f == "ipairs" || f == "pairs") { need(1); Value it; it.t = Value::Func; it.fn = std::make_shared<FuncV>(); it.fn->builtin = 1; it.fn->bname = f + "_iter"; return { it, a[0], Value::num(0) }; }
        if (f == "print") return {};
        if (f == "assert") { need(1); if (!a[0].truthy()) fail(ln + "assertion failed" + (a.size() > 1 && a[1].t == Value::Str ? ": " + a[1].s : "")); return { a[0] }; }
        if (f == "tostring") { need(1); if (a[0].t == Value::Num) { char b[64]; snprintf(b, sizeof b, "%.14g", a[0].n); return { Value::str(b) }; } if (a[0].t == Value::Str) return { a[0] }; return { Value::str("?") }; }
        if (f == "tonumber") { need(1); if (a[0].t == Value::Num) return { a[0] }; if (a[0].t == Value::Str) return { Value::num(atof(a[0].s.c_str())) }; return { Value() }; }
        if (f == "unpack") { need(1); if (a[0].t != Value::Table) fail(ln + "unpack(table)"); return a[0].tab->arr; }
        fail(ln + "builtin " + f + " is not implemented");
    }
    Value decl(const Input& in) { SymV s; s.k = SymV::Image; s.id = (int)P.inputs.size(); P.inputs.push_back(in); return Value::make_sym(s); }
    static E dot(const std::vector<E>& x, const std::vector<E>& y, const std::string& ln)
    { if (x.size() != y.size() || x.empty()) fail(ln + "dot: sizes differ"); E r = bin(Op::Mul, x[0], y[0]); for (size_t i = 1; i < x.size(); ++i) r = bin(Op::Add, r, bin(Op::Mul, x[i], y[i])); return r; }
    static std::vector<E> cross(const std::vector<E>& a, const std::vector<E>& b, const std::string& ln)
    {
        if (a.size() != 3 || b.size() != 3) fail(ln + "cross of 3-vectors");
        auto M = [&](const E& x, const E& y) { return bin(Op::Mul, x, y); };
        return { bin(Op::Sub, M(a[1], b[2]), M(a[2], b[1])), bin(Op::Sub, M(a[2], b[0]), M(a[0], b[2])), bin(Op::Sub, M(a[0], b[1]), M(a[1], b[0])) };
    }
};

}  // namespace

bool run_problem_file(const char* filename, Problem& out, std::string& err, const unsigned* dims)
{
    std::ifstream f(filename, std::ios::binary);
    if (!f) { err = std::string("cannot open ") + filename; return false; }
    std::stringstream ss; ss << f.rdbuf();
    out = Problem(); out.file = filename;
    out.plan_dims = dims;
    try {
        Parser ps(lex(ss.str()));
        NP chunk = ps.block();
        if (ps.cur().k != Tok::End) ps.err("unexpected token");
        Interp in(out);
        try { in.exec_block(chunk, in.globals); } catch (Interp::ReturnEx&) {}
        // drop residuals emptied by merge; validate declarations
        std::vector<Residual> keep; for (auto& r : out.residuals) if (!r.exprs.empty()) keep.push_back(r);
        out.residuals = keep;
        if (out.residuals.empty()) fail("the file defines no Residuals");
        int n_unknown = 0;
        for (auto& in2 : out.inputs) { if (in2.name.empty()) fail("an Unknown / Array / Sparse / Param was created outside Inputs{}"); if (in2.kind == InputKind::Unknown) ++n_unknown; }
        if (!n_unknown) fail("the file declares no Unknown");
        out.plan_dims = nullptr;
        out.dim_alias.resize(out.dims.size(), -1);
    } catch (std::exception& e) { err = std::string(filename) + ": " + e.what(); return false; }
    return true;
}

std::string describe(const Problem& p)
{
    std::ostringstream o;
    o << "dims:"; for (auto& d : p.dims) o << " " << d; o << "\n";
    for (auto& in : p.inputs) {
        o << (in.kind == InputKind::Unknown ? "unknown " : in.kind == InputKind::Array ? "array " : in.kind == InputKind::Sparse ? "sparse " : "param ") << in.name << " slot " << in.slot;
        if (in.kind == InputKind::Unknown || in.kind == InputKind::Array) { o << " channels " << in.channels << (in.is_u8 ? " uint8" : "") << " over"; for (int d : in.dims) o << " " << p.dims[d]; if (in.exclude) o << " (Exclude)"; }
        if (in.kind == InputKind::Sparse) { for (size_t k = 0; k + 1 < in.dims.size(); ++k) o << " " << p.dims[in.dims[k]]; o << " -> " << p.dims[in.dims.back()]; }
        o << "\n";
    }
    o << "preconditioner " << (p.use_preconditioner ? 1 : 0) << "\n";
    if (p.direct_solve) o << "direct_solve\n";
    for (auto& r : p.residuals) { o << "residual " << r.name << " x" << r.exprs.size() << " over"; for (int d : r.domain) o << " " << p.dims[d]; if (r.mat_J) o << " J"; if (r.mat_JtJ) o << " JtJ"; if (r.mat_Jp) o << " Jp"; o << "\n"; }
    for (auto& ca : p.computed) { o << "computed x" << ca.exprs.size() << " over"; for (int d : ca.domain) o << " " << p.dims[d]; o << "\n"; }
    return o.str();
}

}  // namespace dsl
}  // namespace thallo
