// json_min.h -- a minimal JSON reader/writer, enough for Avro schemas (avro.schema metadata of container files).
#pragma once
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mlh {

struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;      // insertion order kept (field order matters for Avro)

    bool is_str() const { return kind == Str; }
    bool is_arr() const { return kind == Arr; }
    bool is_obj() const { return kind == Obj; }
    const Json *get(const std::string &k) const
    {
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

class JsonParser {
  public:
    explicit JsonParser(const std::string &s) : s_(s) {}
    Json parse()
    {
        Json v = value();
        ws();
        if (p_ != s_.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string &s_;
    size_t p_ = 0;
    [[noreturn]] void fail(const char *m) const { throw std::runtime_error(std::string("json: ") + m + " at " + std::to_string(p_)); }
    void ws() { while (p_ < s_.size() && std::isspace((unsigned char)s_[p_])) p_++; }
    Json value()
    {
        ws();
        if (p_ >= s_.size()) fail("eof");
        char c = s_[p_];
        Json v;
        if (c == '{') {
            v.kind = Json::Obj;
            p_++;
            ws();
            if (s_[p_] == '}') { p_++; return v; }
            for (;;) {
                ws();
                Json k = value();
                if (!k.is_str()) fail("object key");
                ws();
                if (s_[p_++] != ':') fail("':'");
                v.obj.emplace_back(k.str, value());
                ws();
                if (s_[p_] == ',') { p_++; continue; }
                if (s_[p_] == '}') { p_++; break; }
                fail("',' or '}'");
            }
        } else if (c == '[') {
            v.kind = Json::Arr;
            p_++;
            ws();
            if (s_[p_] == ']') { p_++; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (s_[p_] == ',') { p_++; continue; }
                if (s_[p_] == ']') { p_++; break; }
                fail("',' or ']'");
            }
        } else if (c == '"') {
            v.kind = Json::Str;
            p_++;
            while (p_ < s_.size() && s_[p_] != '"') {
                if (s_[p_] == '\\') {
                    p_++;
                    char e = s_[p_++];
                    switch (e) {
                    case 'n': v.str += '\n'; break;
                    case 't': v.str += '\t'; break;
                    case 'r': v.str += '\r'; break;
                    case 'b': v.str += '\b'; break;
                    case 'f': v.str += '\f'; break;
                    case 'u': {
                        unsigned cp = (unsigned)std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
                        p_ += 4;
                        if (cp < 0x80) v.str += (char)cp;
                        else if (cp < 0x800) { v.str += (char)(0xC0 | (cp >> 6)); v.str += (char)(0x80 | (cp & 0x3F)); }
                        else { v.str += (char)(0xE0 | (cp >> 12)); v.str += (char)(0x80 | ((cp >> 6) & 0x3F)); v.str += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: v.str += e;
                    }
                } else v.str += s_[p_++];
            }
            p_++;
        } else if (!s_.compare(p_, 4, "true")) { v.kind = Json::Bool; v.b = true; p_ += 4; }
        else if (!s_.compare(p_, 5, "false")) { v.kind = Json::Bool; p_ += 5; }
        else if (!s_.compare(p_, 4, "null")) { p_ += 4; }
        else {
            char *end = nullptr;
            v.kind = Json::Num;
            v.num = std::strtod(s_.c_str() + p_, &end);
            if (end == s_.c_str() + p_) fail("value");
            p_ = (size_t)(end - s_.c_str());
        }
        return v;
    }
};

inline std::string json_escape(const std::string &s)
{
    std::string o = "\"";
    for (char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += c; }
        else if (c == '\n') o += "\\n";
        else if ((unsigned char)c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o += c;
    }
    return o + "\"";
}

// Serialise back to JSON text (schemas only: numbers are printed as integers when integral).
inline void json_dump(const Json &v, std::string &out)
{
    switch (v.kind) {
    case Json::Null: out += "null"; break;
    case Json::Bool: out += v.b ? "true" : "false"; break;
    case Json::Num: {
        char buf[40];
        if (v.num == (double)(long long)v.num) snprintf(buf, sizeof buf, "%lld", (long long)v.num);
        else snprintf(buf, sizeof buf, "%.17g", v.num);
        out += buf;
        break;
    }
    case Json::Str: {
        out += '"';
        for (unsigned char c : v.str) {
            if (c == '"' || c == '\\') { out += '\\'; out += (char)c; }
            else if (c < 0x20) { char buf[8]; snprintf(buf, sizeof buf, "\\u%04x", c); out += buf; }
            else out += (char)c;
        }
        out += '"';
        break;
    }
    case Json::Arr:
        out += '[';
        for (size_t i = 0; i < v.arr.size(); i++) { if (i) out += ','; json_dump(v.arr[i], out); }
        out += ']';
        break;
    case Json::Obj:
        out += '{';
        for (size_t i = 0; i < v.obj.size(); i++) {
            if (i) out += ',';
            Json k; k.kind = Json::Str; k.str = v.obj[i].first;
            json_dump(k, out);
            out += ':';
            json_dump(v.obj[i].second, out);
        }
        out += '}';
        break;
    }
}

}  // namespace mlh
