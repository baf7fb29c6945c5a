// json.hpp — minimal strict JSON DOM for the scene schema (stands in for serde_json,
// reference main.rs:15).  Numbers keep their source text so u64 fields (Texture.width)
// and f64 fields parse the way serde would.  Objects keep insertion order.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rtjson {

struct Value;
using ValuePtr = std::unique_ptr<Value>;

struct Value {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  std::string text;  // Number: source text; String: decoded UTF-8
  std::vector<ValuePtr> items;
  std::vector<std::pair<std::string, ValuePtr>> members;

  const Value* find(const char* key) const {
    for (auto& m : members)
      if (m.first == key) return m.second.get();
    return nullptr;
  }
};

struct ParseError : std::runtime_error {
  size_t offset;
  ParseError(const std::string& m, size_t off)
      : std::runtime_error(m + " at byte " + std::to_string(off)), offset(off) {}
};

class Parser {
 public:
  Parser(const char* s, size_t n) : s_(s), n_(n) {}
  ValuePtr parse() {
    ValuePtr v = value(0);
    ws();
    if (i_ != n_) fail("trailing characters");
    return v;
  }

 private:
  const char* s_;
  size_t n_, i_ = 0;
  [[noreturn]] void fail(const char* m) { throw ParseError(m, i_); }
  void ws() {
    while (i_ < n_ && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) ++i_;
  }
  bool lit(const char* w) {
    size_t l = std::strlen(w);
    if (n_ - i_ >= l && std::memcmp(s_ + i_, w, l) == 0) { i_ += l; return true; }
    return false;
  }
  static void utf8(std::string& out, uint32_t cp) {
    if (cp < 0x80) out += char(cp);
    else if (cp < 0x800) { out += char(0xC0 | (cp >> 6)); out += char(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) {
      out += char(0xE0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F));
    } else {
      out += char(0xF0 | (cp >> 18)); out += char(0x80 | ((cp >> 12) & 0x3F));
      out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F));
    }
  }
  uint32_t hex4() {
    if (n_ - i_ < 4) fail("bad \\u escape");
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) {
      char c = s_[i_++];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= c - '0';
      else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
      else fail("bad \\u escape");
    }
    return v;
  }
  std::string string() {
    if (i_ >= n_ || s_[i_] != '"') fail("expected string");
    ++i_;
    std::string out;
    while (true) {
      if (i_ >= n_) fail("unterminated string");
      unsigned char c = s_[i_++];
      if (c == '"') break;
      if (c < 0x20) fail("control character in string");
      if (c != '\\') { out += char(c); continue; }
      if (i_ >= n_) fail("unterminated escape");
      char e = s_[i_++];
      switch (e) {
        case '"': out += '"'; break;
        case '\\': out += '\\'; break;
        case '/': out += '/'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'n': out += '\n'; break;
        case 'r': out += '\r'; break;
        case 't': out += '\t'; break;
        case 'u': {
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp < 0xDC00) {
            if (!(n_ - i_ >= 2 && s_[i_] == '\\' && s_[i_ + 1] == 'u')) fail("lone surrogate");
            i_ += 2;
            uint32_t lo = hex4();
            if (lo < 0xDC00 || lo > 0xDFFF) fail("bad surrogate pair");
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          } else if (cp >= 0xDC00 && cp <= 0xDFFF) fail("lone surrogate");
          utf8(out, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
    return out;
  }
  std::string number() {
    size_t st = i_;
    if (i_ < n_ && s_[i_] == '-') ++i_;
    if (i_ >= n_) fail("bad number");
    if (s_[i_] == '0') ++i_;
    else if (s_[i_] >= '1' && s_[i_] <= '9') { while (i_ < n_ && s_[i_] >= '0' && s_[i_] <= '9') ++i_; }
    else fail("bad number");
    if (i_ < n_ && s_[i_] == '.') {
      ++i_;
      if (!(i_ < n_ && s_[i_] >= '0' && s_[i_] <= '9')) fail("bad fraction");
      while (i_ < n_ && s_[i_] >= '0' && s_[i_] <= '9') ++i_;
    }
    if (i_ < n_ && (s_[i_] == 'e' || s_[i_] == 'E')) {
      ++i_;
      if (i_ < n_ && (s_[i_] == '+' || s_[i_] == '-')) ++i_;
      if (!(i_ < n_ && s_[i_] >= '0' && s_[i_] <= '9')) fail("bad exponent");
      while (i_ < n_ && s_[i_] >= '0' && s_[i_] <= '9') ++i_;
    }
    return std::string(s_ + st, i_ - st);
  }
  ValuePtr value(int depth) {
    if (depth > 128) fail("nesting too deep");  // serde_json's recursion limit
    ws();
    if (i_ >= n_) fail("unexpected end of input");
    ValuePtr v(new Value);
    char c = s_[i_];
    if (c == '{') {
      ++i_;
      v->kind = Value::Object;
      ws();
      if (i_ < n_ && s_[i_] == '}') { ++i_; return v; }
      while (true) {
        ws();
        std::string k = string();
        ws();
        if (i_ >= n_ || s_[i_] != ':') fail("expected ':'");
        ++i_;
        v->members.emplace_back(std::move(k), value(depth + 1));
        ws();
        if (i_ < n_ && s_[i_] == ',') { ++i_; continue; }
        if (i_ < n_ && s_[i_] == '}') { ++i_; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      ++i_;
      v->kind = Value::Array;
      ws();
      if (i_ < n_ && s_[i_] == ']') { ++i_; return v; }
      while (true) {
        v->items.push_back(value(depth + 1));
        ws();
        if (i_ < n_ && s_[i_] == ',') { ++i_; continue; }
        if (i_ < n_ && s_[i_] == ']') { ++i_; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      v->kind = Value::String;
      v->text = string();
    } else if (c == 't') {
      if (!lit("true")) fail("bad literal");
      v->kind = Value::Bool; v->b = true;
    } else if (c == 'f') {
      if (!lit("false")) fail("bad literal");
      v->kind = Value::Bool; v->b = false;
    } else if (c == 'n') {
      if (!lit("null")) fail("bad literal");
      v->kind = Value::Null;
    } else {
      v->kind = Value::Number;
      v->text = number();
    }
    return v;
  }
};

inline ValuePtr parse(const char* s, size_t n) { return Parser(s, n).parse(); }

}  // namespace rtjson
