#include "cartslam_amd/json.hpp"

#include <cctype>

namespace cart::json {
namespace {
struct Parser {
    const std::string &s;
    size_t i = 0;
    int depth = 0;   // open arrays / objects: bounded, so that a hostile file cannot overflow the stack (found by fuzz_readers under ASan)
    static constexpr int kMaxDepth = 64;
    struct Nest {
        Parser &p;
        explicit Nest(Parser &p) : p(p) { if (++p.depth > kMaxDepth) p.err("nesting deeper than 64 levels"); }
        ~Nest() { --p.depth; }
    };
    explicit Parser(const std::string &s) : s(s) {}
    [[noreturn]] void err(const std::string &m) { throw std::runtime_error("JSON parse error at offset " + std::to_string(i) + ": " + m); }
    void ws() { while (i < s.size() && std::isspace((unsigned char)s[i])) ++i; }
    Value value() {
        ws();
        if (i >= s.size()) err("unexpected end");
        const char c = s[i];
        Value v;
        if (c == '{') {
            Nest nest(*this);
            v.kind = Value::Object; ++i; ws();
            if (i < s.size() && s[i] == '}') { ++i; return v; }
            for (;;) {
                ws();
                if (i >= s.size() || s[i] != '"') err("expected string key");
                std::string k = str();
                ws();
                if (i >= s.size() || s[i] != ':') err("expected ':'");
                ++i;
                v.obj[k] = value();
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; return v; }
                err("expected ',' or '}'");
            }
        }
        if (c == '[') {
            Nest nest(*this);
            v.kind = Value::Array; ++i; ws();
            if (i < s.size() && s[i] == ']') { ++i; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; return v; }
                err("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = Value::String; v.str = str(); return v; }
        if (s.compare(i, 4, "true") == 0) { v.kind = Value::Bool; v.b = true; i += 4; return v; }
        if (s.compare(i, 5, "false") == 0) { v.kind = Value::Bool; v.b = false; i += 5; return v; }
        if (s.compare(i, 4, "null") == 0) { i += 4; return v; }
        char *end = nullptr;
        v.num = std::strtod(s.c_str() + i, &end);
        if (end == s.c_str() + i) err("unexpected character");
        v.kind = Value::Number;
        i = (size_t)(end - s.c_str());
        return v;
    }
    std::string str() {
        std::string out;
        ++i;  // opening quote
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\' && i + 1 < s.size()) {
                const char e = s[i + 1];
                out += e == 'n' ? '\n' : e == 't' ? '\t' : e;
                i += 2;
            } else out += s[i++];
        }
        if (i >= s.size()) err("unterminated string");
        ++i;
        return out;
    }
};
}  // namespace

Value parse(const std::string &text) {
    Parser p(text);
    Value v = p.value();
    p.ws();
    if (p.i != text.size()) p.err("trailing characters");
    return v;
}
}  // namespace cart::json
