// mini_json.h -- a small recursive-descent JSON reader (objects, arrays, numbers, strings, bools,
// null).  Replaces the reference's vendored nlohmann json.hpp for the two files the headless path
// parses: options/opt.json (render_options.hpp:61-77) and blender transforms_*.json
// (main_headless.cpp:255-272).
#pragma once
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rto {
namespace json {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<ValuePtr> arr;
    std::map<std::string, ValuePtr> obj;

    bool has(const std::string& k) const { return kind == Object && obj.count(k) != 0; }
    const Value& at(const std::string& k) const {
        if (kind != Object) throw std::runtime_error("json: not an object (key '" + k + "')");
        auto it = obj.find(k);
        if (it == obj.end()) throw std::runtime_error("json: key '" + k + "' not found");
        return *it->second;
    }
    const Value& at(size_t i) const {
        if (kind != Array || i >= arr.size()) throw std::runtime_error("json: bad array index");
        return *arr[i];
    }
    size_t size() const { return kind == Array ? arr.size() : (kind == Object ? obj.size() : 0); }
    double as_number() const {
        if (kind != Number) throw std::runtime_error("json: number expected");
        return num;
    }
    // nlohmann converts numbers <-> bools leniently for arithmetic targets; opt.json uses real bools
    bool as_bool() const {
        if (kind == Bool) return b;
        if (kind == Number) return num != 0;
        throw std::runtime_error("json: bool expected");
    }
};

class Parser {
public:
    explicit Parser(const std::string& s) : s_(s) {}
    ValuePtr parse() {
        ValuePtr v = value();
        ws();
        if (p_ != s_.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& s_;
    size_t p_ = 0;
    int depth_ = 0;  // of containers being parsed: bounded, the parser recurses per level
    static constexpr int kMaxDepth = 256;
    struct Nest {
        Parser& p;
        explicit Nest(Parser& q) : p(q) {
            if (++p.depth_ > kMaxDepth) p.fail("nested too deeply");
        }
        ~Nest() { --p.depth_; }
    };

    [[noreturn]] void fail(const char* what) const {
        throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(p_));
    }
    void ws() {
        while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\t' || s_[p_] == '\n' || s_[p_] == '\r')) ++p_;
    }
    bool lit(const char* t) {
        size_t n = 0;
        while (t[n]) ++n;
        if (s_.compare(p_, n, t) == 0) {
            p_ += n;
            return true;
        }
        return false;
    }
    ValuePtr value() {
        ws();
        if (p_ >= s_.size()) fail("unexpected end");
        auto v = std::make_shared<Value>();
        const char c = s_[p_];
        if (c == '{') {
            const Nest nest(*this);
            v->kind = Value::Object;
            ++p_;
            ws();
            if (p_ < s_.size() && s_[p_] == '}') {
                ++p_;
                return v;
            }
            for (;;) {
                ws();
                if (p_ >= s_.size() || s_[p_] != '"') fail("object key expected");
                std::string k = string();
                ws();
                if (p_ >= s_.size() || s_[p_] != ':') fail("':' expected");
                ++p_;
                v->obj[k] = value();
                ws();
                if (p_ < s_.size() && s_[p_] == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < s_.size() && s_[p_] == '}') {
                    ++p_;
                    return v;
                }
                fail("',' or '}' expected");
            }
        } else if (c == '[') {
            const Nest nest(*this);
            v->kind = Value::Array;
            ++p_;
            ws();
            if (p_ < s_.size() && s_[p_] == ']') {
                ++p_;
                return v;
            }
            for (;;) {
                v->arr.push_back(value());
                ws();
                if (p_ < s_.size() && s_[p_] == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < s_.size() && s_[p_] == ']') {
                    ++p_;
                    return v;
                }
                fail("',' or ']' expected");
            }
        } else if (c == '"') {
            v->kind = Value::String;
            v->str = string();
            return v;
        } else if (lit("true")) {
            v->kind = Value::Bool;
            v->b = true;
            return v;
        } else if (lit("false")) {
            v->kind = Value::Bool;
            v->b = false;
            return v;
        } else if (lit("null")) {
            return v;
        } else {
            const char* start = s_.c_str() + p_;
            char* end = nullptr;
            const double d = std::strtod(start, &end);
            if (end == start) fail("value expected");
            p_ += (size_t)(end - start);
            v->kind = Value::Number;
            v->num = d;
            return v;
        }
    }
    std::string string() {
        std::string out;
        ++p_;  // opening quote
        while (p_ < s_.size() && s_[p_] != '"') {
            char c = s_[p_++];
            if (c == '\\') {
                if (p_ >= s_.size()) fail("bad escape");
                const char e = s_[p_++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (p_ + 4 > s_.size()) fail("bad \\u escape");
                        const unsigned cp = (unsigned)std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
                        p_ += 4;
                        if (cp < 0x80) {
                            out += (char)cp;
                        } else if (cp < 0x800) {
                            out += (char)(0xC0 | (cp >> 6));
                            out += (char)(0x80 | (cp & 0x3F));
                        } else {
                            out += (char)(0xE0 | (cp >> 12));
                            out += (char)(0x80 | ((cp >> 6) & 0x3F));
                            out += (char)(0x80 | (cp & 0x3F));
                        }
                        break;
                    }
                    default: out += e;
                }
            } else {
                out += c;
            }
        }
        if (p_ >= s_.size()) fail("unterminated string");
        ++p_;
        return out;
    }
};

inline ValuePtr parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace json
}  // namespace rto
