// Minimal JSON reader for config.json / generation_config.json / safetensors headers (the
// reference uses serde_json: qwen3/generate.rs:25-26,33-35).  Recursive descent over the whole RFC 8259 grammar;
// numbers are kept as double plus the original text (safetensors offsets exceed 2^53 only in theory, but integers are
// re-parsed with strtoull to stay exact).
#pragma once
#include <stdlib.h>
#include <string.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace aha {

struct JsonValue {
  enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
  bool b = false;
  double num = 0;
  std::string str;  // STR: decoded text; NUM: the literal
  std::vector<JsonValue> arr;
  std::vector<std::pair<std::string, JsonValue>> obj;  // insertion order (safetensors headers are large: no map copy)

  const JsonValue* get(const char* key) const {
    if (kind != OBJ) return nullptr;
    for (const auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  bool is_num() const { return kind == NUM; }
  int64_t as_i64() const { return kind == NUM ? (int64_t)strtoll(str.c_str(), nullptr, 10) : 0; }
  uint64_t as_u64() const { return kind == NUM ? (uint64_t)strtoull(str.c_str(), nullptr, 10) : 0; }
};

class JsonParser {
 public:
  JsonParser(const char* p, size_t n) : p_(p), end_(p + n) {}
  // returns false and fills err on malformed input
  bool parse(JsonValue* out, std::string* err) {
    skip_ws();
    if (!value(out, 0)) {
      *err = err_.empty() ? std::string("malformed JSON") : err_;
      return false;
    }
    skip_ws();
    if (p_ != end_) {
      *err = "trailing characters after the JSON value";
      return false;
    }
    return true;
  }

 private:
  const char* p_;
  const char* end_;
  std::string err_;

  void skip_ws() {
    while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
  }
  bool fail(const char* m) {
    if (err_.empty()) err_ = m;
    return false;
  }
  bool lit(const char* s) {
    const size_t n = strlen(s);
    if ((size_t)(end_ - p_) < n || memcmp(p_, s, n) != 0) return fail("bad literal");
    p_ += n;
    return true;
  }
  static void utf8(std::string* s, unsigned cp) {
    if (cp < 0x80) {
      s->push_back((char)cp);
    } else if (cp < 0x800) {
      s->push_back((char)(0xC0 | (cp >> 6)));
      s->push_back((char)(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
      s->push_back((char)(0xE0 | (cp >> 12)));
      s->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      s->push_back((char)(0x80 | (cp & 0x3F)));
    } else {
      s->push_back((char)(0xF0 | (cp >> 18)));
      s->push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
      s->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      s->push_back((char)(0x80 | (cp & 0x3F)));
    }
  }
  bool hex4(unsigned* v) {
    if (end_ - p_ < 4) return fail("short \\u escape");
    unsigned x = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = *p_++;
      x <<= 4;
      if (c >= '0' && c <= '9') x |= (unsigned)(c - '0');
      else if (c >= 'a' && c <= 'f') x |= (unsigned)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') x |= (unsigned)(c - 'A' + 10);
      else return fail("bad \\u escape");
    }
    *v = x;
    return true;
  }
  bool string(std::string* s) {
    if (p_ >= end_ || *p_ != '"') return fail("expected string");
    ++p_;
    while (p_ < end_) {
      const char c = *p_++;
      if (c == '"') return true;
      if (c == '\\') {
        if (p_ >= end_) break;
        const char e = *p_++;
        switch (e) {
          case '"': s->push_back('"'); break;
          case '\\': s->push_back('\\'); break;
          case '/': s->push_back('/'); break;
          case 'b': s->push_back('\b'); break;
          case 'f': s->push_back('\f'); break;
          case 'n': s->push_back('\n'); break;
          case 'r': s->push_back('\r'); break;
          case 't': s->push_back('\t'); break;
          case 'u': {
            unsigned cp;
            if (!hex4(&cp)) return false;
            if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
              p_ += 2;
              unsigned lo;
              if (!hex4(&lo)) return false;
              cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
            }
            utf8(s, cp);
            break;
          }
          default: return fail("bad escape");
        }
      } else {
        s->push_back(c);
      }
    }
    return fail("unterminated string");
  }
  bool number(JsonValue* v) {
    const char* s = p_;
    if (p_ < end_ && *p_ == '-') ++p_;
    while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '+' || *p_ == '-')) ++p_;
    if (p_ == s) return fail("expected a value");
    v->kind = JsonValue::NUM;
    v->str.assign(s, (size_t)(p_ - s));
    char* endp = nullptr;
    v->num = strtod(v->str.c_str(), &endp);
    if (endp == v->str.c_str() || *endp != '\0') return fail("malformed number");
    return true;
  }
  bool value(JsonValue* v, int depth) {
    if (depth > 64) return fail("nesting too deep");
    skip_ws();
    if (p_ >= end_) return fail("unexpected end of input");
    const char c = *p_;
    if (c == '{') {
      ++p_;
      v->kind = JsonValue::OBJ;
      skip_ws();
      if (p_ < end_ && *p_ == '}') { ++p_; return true; }
      for (;;) {
        skip_ws();
        std::string key;
        if (!string(&key)) return false;
        skip_ws();
        if (p_ >= end_ || *p_ != ':') return fail("expected ':'");
        ++p_;
        v->obj.emplace_back(std::move(key), JsonValue());
        if (!value(&v->obj.back().second, depth + 1)) return false;
        skip_ws();
        if (p_ < end_ && *p_ == ',') { ++p_; continue; }
        if (p_ < end_ && *p_ == '}') { ++p_; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      ++p_;
      v->kind = JsonValue::ARR;
      skip_ws();
      if (p_ < end_ && *p_ == ']') { ++p_; return true; }
      for (;;) {
        v->arr.emplace_back();
        if (!value(&v->arr.back(), depth + 1)) return false;
        skip_ws();
        if (p_ < end_ && *p_ == ',') { ++p_; continue; }
        if (p_ < end_ && *p_ == ']') { ++p_; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') {
      v->kind = JsonValue::STR;
      return string(&v->str);
    }
    if (c == 't') { v->kind = JsonValue::BOOL; v->b = true; return lit("true"); }
    if (c == 'f') { v->kind = JsonValue::BOOL; v->b = false; return lit("false"); }
    if (c == 'n') { v->kind = JsonValue::NUL; return lit("null"); }
    return number(v);
  }
};

}  // namespace aha
