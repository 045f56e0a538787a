// TEST INFRASTRUCTURE: a small JSON reader behind the part of json11's interface the reference's C++ uses (see README.md):
// Json::parse(text, err), operator[](key), int_value / number_value / bool_value / string_value with json11's defaults for a
// missing key or a value of another type (0, 0.0, false, "").
#pragma once
#include <cctype>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace json11 {

class Json {
 public:
    enum Type { NUL, NUMBER, BOOL, STRING, ARRAY, OBJECT };
    Json() {}
    Type type() const { return v_ ? v_->type : NUL; }
    double number_value() const { return type() == NUMBER ? v_->num : 0.0; }
    int int_value() const { return type() == NUMBER ? static_cast<int>(v_->num) : 0; }
    bool bool_value() const { return type() == BOOL ? v_->flag : false; }
    const std::string& string_value() const {
        static const std::string empty;
        return type() == STRING ? v_->str : empty;
    }
    const Json& operator[](const std::string& key) const {
        static const Json none;
        if (type() != OBJECT) return none;
        auto it = v_->obj.find(key);
        return it == v_->obj.end() ? none : it->second;
    }
    static Json parse(const std::string& in, std::string& err) {
        size_t at = 0;
        Json j = value(in, at, err);
        if (err.empty()) {
            skip(in, at);
            if (at != in.size()) err = "unexpected trailing text";
        }
        return err.empty() ? j : Json();
    }

 private:
    struct Node {
        Type type = NUL;
        double num = 0.0;
        bool flag = false;
        std::string str;
        std::vector<Json> arr;
        std::map<std::string, Json> obj;
    };
    std::shared_ptr<Node> v_;

    static void skip(const std::string& s, size_t& at) {
        while (at < s.size() && std::isspace(static_cast<unsigned char>(s[at]))) ++at;
    }
    static std::string text(const std::string& s, size_t& at, std::string& err) {
        std::string out;
        ++at;  // opening quote
        while (at < s.size() && s[at] != '"') {
            if (s[at] == '\\' && at + 1 < s.size()) {
                const char c = s[++at];
                if (c == 'n') out += '\n';
                else if (c == 't') out += '\t';
                else if (c == 'u') { out += '?'; at += 4; }
                else out += c;
            } else {
                out += s[at];
            }
            ++at;
        }
        if (at >= s.size()) err = "unterminated string";
        ++at;
        return out;
    }
    static Json value(const std::string& s, size_t& at, std::string& err) {
        skip(s, at);
        Json j;
        j.v_ = std::make_shared<Node>();
        if (at >= s.size()) { err = "unexpected end"; return j; }
        const char c = s[at];
        if (c == '{') {
            j.v_->type = OBJECT;
            ++at;
            skip(s, at);
            if (at < s.size() && s[at] == '}') { ++at; return j; }
            while (err.empty()) {
                skip(s, at);
                if (at >= s.size() || s[at] != '"') { err = "expected a key"; break; }
                const std::string key = text(s, at, err);
                skip(s, at);
                if (at >= s.size() || s[at] != ':') { err = "expected ':'"; break; }
                ++at;
                j.v_->obj[key] = value(s, at, err);
                skip(s, at);
                if (at < s.size() && s[at] == ',') { ++at; continue; }
                if (at < s.size() && s[at] == '}') { ++at; break; }
                err = "expected ',' or '}'";
            }
        } else if (c == '[') {
            j.v_->type = ARRAY;
            ++at;
            skip(s, at);
            if (at < s.size() && s[at] == ']') { ++at; return j; }
            while (err.empty()) {
                j.v_->arr.push_back(value(s, at, err));
                skip(s, at);
                if (at < s.size() && s[at] == ',') { ++at; continue; }
                if (at < s.size() && s[at] == ']') { ++at; break; }
                err = "expected ',' or ']'";
            }
        } else if (c == '"') {
            j.v_->type = STRING;
            j.v_->str = text(s, at, err);
        } else if (s.compare(at, 4, "true") == 0) {
            j.v_->type = BOOL; j.v_->flag = true; at += 4;
        } else if (s.compare(at, 5, "false") == 0) {
            j.v_->type = BOOL; j.v_->flag = false; at += 5;
        } else if (s.compare(at, 4, "null") == 0) {
            at += 4;
        } else {
            char* end = nullptr;
            j.v_->num = std::strtod(s.c_str() + at, &end);
            if (end == s.c_str() + at) { err = "unexpected character"; return j; }
            j.v_->type = NUMBER;
            at = static_cast<size_t>(end - s.c_str());
        }
        return j;
    }
};

}  // namespace json11
