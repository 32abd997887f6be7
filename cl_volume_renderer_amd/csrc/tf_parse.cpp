// tf_parse.cpp -- turns the reference's generated transfer-function source into a rule table.
//
// The reference prepends an OpenCL-C function to the kernel source and JIT-compiles it
// (app/ui.cpp:160-168 builds it from app/tf_part.cpp:55-79; tests/sdf/sdf_test.cpp:22 and
// app/sdf_benchmark.cpp:18 use a hand-written one-liner).  Here the same text is parsed once per
// clwh_kernel_get and becomes a launch-time parameter, so a TF flush never recompiles anything.
//
// Grammar accepted:
//   inline bool is_event_gen(short value, short gradient, (u)int4 *color) { stmt* }
//   stmt  := if ( cond ) { int4 tmp_color = {r,g,b,a}; *color = tmp_color; return true; }
//          | return false; | return true; | return cond;
//   cond  := term ( && term )*
//   term  := ( cond ) | (value|gradient) (>=|<=|>|<|==) number
// `value` and `gradient` are shorts compared against decimal literals, i.e. integers against
// reals; each comparison is folded into inclusive integer bounds.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/clwh.h"

namespace {

struct Tok {
  enum Kind { Ident, Number, Sym, End } kind;
  std::string text;
  double num = 0.0;
};

class Lexer {
 public:
  explicit Lexer(const char *s) : p_(s) {}
  bool run(std::vector<Tok> &out) {
    while (*p_) {
      if (std::isspace((unsigned char)*p_)) { ++p_; continue; }
      if (std::isalpha((unsigned char)*p_) || *p_ == '_') {
        const char *b = p_;
        while (std::isalnum((unsigned char)*p_) || *p_ == '_') ++p_;
        out.push_back({Tok::Ident, std::string(b, p_), 0.0});
        continue;
      }
      if (std::isdigit((unsigned char)*p_) || (*p_ == '.' && std::isdigit((unsigned char)p_[1]))) {
        char *e = nullptr;
        double v = std::strtod(p_, &e);
        if (e == p_) return false;
        p_ = e;
        if (*p_ == 'f' || *p_ == 'F') ++p_;
        out.push_back({Tok::Number, "", v});
        continue;
      }
      static const char *two[] = {">=", "<=", "==", "&&", "!=", "||"};
      bool matched = false;
      for (const char *t : two)
        if (p_[0] == t[0] && p_[1] == t[1]) {
          out.push_back({Tok::Sym, t, 0.0});
          p_ += 2;
          matched = true;
          break;
        }
      if (matched) continue;
      out.push_back({Tok::Sym, std::string(1, *p_), 0.0});
      ++p_;
    }
    out.push_back({Tok::End, "", 0.0});
    return true;
  }

 private:
  const char *p_;
};

constexpr int kShortMin = -32768, kShortMax = 32767;

struct Bounds {
  long v_lo = kShortMin, v_hi = kShortMax, g_lo = kShortMin, g_hi = kShortMax;
  bool uses_g = false;
};

class Parser {
 public:
  explicit Parser(const std::vector<Tok> &t) : t_(t) {}

  bool parse(clwh_tf *out) {
    std::memset(out, 0, sizeof *out);
    // header: everything up to the opening brace of the function body
    while (!is_sym("{")) {
      if (at_end()) return false;
      ++i_;
    }
    ++i_;
    while (!is_sym("}")) {
      if (at_end()) return false;
      if (out->n >= CLWH_TF_MAX_RULES) return false;
      if (is_ident("if")) {
        ++i_;
        Bounds b;
        if (!expect("(") || !cond(b) || !expect(")")) return false;
        clwh_tf_rule &r = out->rules[out->n];
        if (!color_block(r)) return false;
        store(r, b, /*writes_color=*/1, /*terminal=*/0);
        out->n++;
      } else if (is_ident("return")) {
        ++i_;
        if (is_ident("false")) {
          ++i_;
          if (!expect(";")) return false;
          skip_to_close();
          return true;
        }
        Bounds b;
        if (is_ident("true")) {
          ++i_;
        } else if (!cond(b)) {
          return false;
        }
        if (!expect(";")) return false;
        store(out->rules[out->n], b, /*writes_color=*/0, /*terminal=*/1);
        out->n++;
        skip_to_close();
        return true;
      } else {
        return false;
      }
    }
    return true;
  }

 private:
  const std::vector<Tok> &t_;
  size_t i_ = 0;

  bool at_end() const { return t_[i_].kind == Tok::End; }
  bool is_sym(const char *s) const { return t_[i_].kind == Tok::Sym && t_[i_].text == s; }
  bool is_ident(const char *s) const { return t_[i_].kind == Tok::Ident && t_[i_].text == s; }
  bool expect(const char *s) {
    if (!is_sym(s)) return false;
    ++i_;
    return true;
  }
  void skip_to_close() {
    while (!at_end() && !is_sym("}")) ++i_;
  }

  static void store(clwh_tf_rule &r, const Bounds &b, int writes_color, int terminal) {
    r.v_lo = (int32_t)b.v_lo; r.v_hi = (int32_t)b.v_hi;
    r.g_lo = (int32_t)b.g_lo; r.g_hi = (int32_t)b.g_hi;
    r.use_gradient = b.uses_g ? 1 : 0;
    r.writes_color = writes_color;
    r.terminal = terminal;
  }

  bool number(double &v) {
    double sign = 1.0;
    while (is_sym("-") || is_sym("+")) {
      if (is_sym("-")) sign = -sign;
      ++i_;
    }
    if (t_[i_].kind != Tok::Number) return false;
    v = sign * t_[i_].num;
    ++i_;
    return true;
  }

  bool term(Bounds &b) {
    if (is_sym("(")) {
      ++i_;
      return cond(b) && expect(")");
    }
    bool on_gradient;
    if (is_ident("value")) on_gradient = false;
    else if (is_ident("gradient")) on_gradient = true;
    else return false;
    ++i_;
    if (t_[i_].kind != Tok::Sym) return false;
    const std::string op = t_[i_].text;
    ++i_;
    double x;
    if (!number(x)) return false;
    long &lo = on_gradient ? b.g_lo : b.v_lo;
    long &hi = on_gradient ? b.g_hi : b.v_hi;
    b.uses_g |= on_gradient;
    // clamp the literal so that the long conversions below cannot overflow
    if (x > 1e9) x = 1e9;
    if (x < -1e9) x = -1e9;
    if (op == ">=") lo = std::max(lo, (long)std::ceil(x));
    else if (op == ">") lo = std::max(lo, (long)std::floor(x) + 1);
    else if (op == "<=") hi = std::min(hi, (long)std::floor(x));
    else if (op == "<") hi = std::min(hi, (long)std::ceil(x) - 1);
    else if (op == "==") {
      if (x == std::floor(x)) { lo = std::max(lo, (long)x); hi = std::min(hi, (long)x); }
      else { lo = 1; hi = 0; }
    } else return false;
    lo = std::max(lo, (long)kShortMin);
    hi = std::min(hi, (long)kShortMax);
    return true;
  }

  bool cond(Bounds &b) {
    if (!term(b)) return false;
    while (is_sym("&&")) {
      ++i_;
      if (!term(b)) return false;
    }
    return true;
  }

  // { int4 tmp_color = {r,g,b,a}; *color = tmp_color; return true; }
  bool color_block(clwh_tf_rule &r) {
    if (!expect("{")) return false;
    if (!(is_ident("int4") || is_ident("uint4"))) return false;
    ++i_;
    if (t_[i_].kind != Tok::Ident) return false;
    const std::string tmp = t_[i_].text;
    ++i_;
    if (!expect("=") || !expect("{")) return false;
    for (int k = 0; k < 4; ++k) {
      double v;
      if (!number(v)) return false;
      r.color[k] = (int32_t)v;
      if (k < 3 && !expect(",")) return false;
    }
    if (!expect("}") || !expect(";")) return false;
    if (!expect("*") || !is_ident("color")) return false;
    ++i_;
    if (!expect("=") || !is_ident(tmp.c_str())) return false;
    ++i_;
    if (!expect(";")) return false;
    if (!is_ident("return")) return false;
    ++i_;
    if (!is_ident("true")) return false;
    ++i_;
    return expect(";") && expect("}");
  }
};

}  // namespace

extern "C" int clwh_tf_parse(const char *source, clwh_tf *out) {
  if (!source || !out) return CLWH_ERR_INVALID_VALUE;
  std::vector<Tok> toks;
  Lexer lx(source);
  if (!lx.run(toks)) return CLWH_ERR_TF_UNSUPPORTED;
  Parser ps(toks);
  if (!ps.parse(out)) {
    std::memset(out, 0, sizeof *out);
    return CLWH_ERR_TF_UNSUPPORTED;
  }
  return CLWH_OK;
}
