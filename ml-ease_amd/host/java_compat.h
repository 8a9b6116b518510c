// java_compat.h -- the few JDK behaviours the hot path's numbers pass through:
//   Float.toString / String.valueOf(float)  (model keys "1.0", liblinear epsilon; jobs/RegressionAdmmTrain.java:184,346,650,702)
//   Double.parseDouble(String.valueOf(float)) round trip (utils/Util.java:145-155)
//   java.util.Properties .job files with typed getters and the reference's defaults (mapred/JobConfig.java:78-243)
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace mlh {

// Shortest decimal digits (at least two significant) that round-trip the float32, as Float.toString prints them.
inline void float_shortest_digits(float f, std::string &digits, int &exp10)
{
    char buf[64];
    for (int prec = 2; prec <= 9; prec++) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, (double)f);
        if (strtof(buf, nullptr) == f) break;
    }
    // buf = d.ddddde[+-]xx
    std::string s(buf);
    size_t e = s.find('e');
    std::string mant = s.substr(0, e);
    exp10 = atoi(s.c_str() + e + 1);
    digits.clear();
    for (char c : mant) if (c >= '0' && c <= '9') digits += c;
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
}

inline std::string java_float_to_string(float f)
{
    if (std::isnan(f)) return "NaN";
    if (std::isinf(f)) return f > 0 ? "Infinity" : "-Infinity";
    if (f == 0) return std::signbit(f) ? "-0.0" : "0.0";
    std::string sign = f < 0 ? "-" : "";
    std::string d;
    int e;
    float_shortest_digits(std::fabs(f), d, e);
    const double a = std::fabs((double)f);
    if (a >= 1e-3 && a < 1e7) {
        if (e >= 0) {
            if ((int)d.size() <= e + 1) return sign + d + std::string((size_t)(e + 1 - (int)d.size()), '0') + ".0";
            return sign + d.substr(0, (size_t)e + 1) + "." + d.substr((size_t)e + 1);
        }
        return sign + "0." + std::string((size_t)(-e - 1), '0') + d;
    }
    return sign + d.substr(0, 1) + "." + (d.size() > 1 ? d.substr(1) : std::string("0")) + "E" + std::to_string(e);
}

inline double float_string_roundtrip(float f) { return strtod(java_float_to_string(f).c_str(), nullptr); }

class JobConfig {
  public:
    static JobConfig from_file(const std::string &path)
    {
        std::ifstream in(path);
        if (!in) throw std::runtime_error("cannot open job file " + path);
        JobConfig c;
        std::string line;
        while (std::getline(in, line)) {
            size_t b = line.find_first_not_of(" \t\r\f");
            if (b == std::string::npos) continue;
            if (line[b] == '#' || line[b] == '!') continue;
            size_t sep = line.find_first_of("=:", b);
            std::string k = trim(line.substr(b, sep == std::string::npos ? std::string::npos : sep - b));
            std::string v = sep == std::string::npos ? "" : trim(line.substr(sep + 1));
            c.props_[k] = v;
        }
        return c;
    }
    bool has(const std::string &k) const { return props_.count(k) > 0; }
    void put(const std::string &k, const std::string &v) { props_[k] = v; }
    std::string get_string(const std::string &k) const
    {
        auto it = props_.find(k);
        if (it == props_.end()) throw std::runtime_error("Undefined property: " + k);      // UndefinedPropertyException
        return it->second;
    }
    std::string get_string(const std::string &k, const std::string &d) const { return has(k) ? props_.at(k) : d; }
    int get_int(const std::string &k) const { return atoi(get_string(k).c_str()); }
    int get_int(const std::string &k, int d) const { return has(k) ? atoi(props_.at(k).c_str()) : d; }
    long get_long(const std::string &k, long d) const { return has(k) ? atol(props_.at(k).c_str()) : d; }
    double get_double(const std::string &k, double d) const { return has(k) ? strtod(props_.at(k).c_str(), nullptr) : d; }
    float get_float(const std::string &k, float d) const { return has(k) ? strtof(props_.at(k).c_str(), nullptr) : d; }
    bool get_bool(const std::string &k, bool d) const
    {
        if (!has(k)) return d;
        std::string v = props_.at(k);
        for (auto &ch : v) ch = (char)tolower(ch);
        return v == "true";                                                                 // Boolean.parseBoolean
    }
    std::vector<std::string> get_list(const std::string &k, char sep) const
    {
        std::vector<std::string> out;
        if (!has(k)) return out;
        std::string v = props_.at(k), cur;
        for (char c : v) {
            if (c == sep) { out.push_back(trim(cur)); cur.clear(); }
            else cur += c;
        }
        if (!trim(cur).empty() || !out.empty()) out.push_back(trim(cur));
        return out;
    }

  private:
    std::map<std::string, std::string> props_;
    static std::string trim(const std::string &s)
    {
        size_t b = s.find_first_not_of(" \t\r\n\f"), e = s.find_last_not_of(" \t\r\n\f");
        return b == std::string::npos ? "" : s.substr(b, e - b + 1);
    }
};

}  // namespace mlh
