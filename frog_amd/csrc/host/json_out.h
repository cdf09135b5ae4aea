// json_out.h -- minimal JSON writer with picojson's number formatting.
//
// The reference serialises its outputs with picojson (kazuho/picojson, pulled in
// through the un-vendored vtkOpenSURF3D submodule, imageGroup.h:8).  picojson
// prints a number as "%.f" when it is integral and below 2^53, else "%.17g", and
// emits object keys in std::map (sorted) order; this writer does the same so the
// files read back identically.  Consumers compare numerically, not textually.
#pragma once

#include <cmath>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

namespace frogjson {

struct Value {
    enum Kind { Null, Number, String, Array, Object } kind = Null;
    double num = 0;
    std::string str;
    std::vector<Value> arr;
    std::map<std::string, Value> obj;

    Value() {}
    Value(double d) : kind(Number), num(d) {}
    Value(const std::string &s) : kind(String), str(s) {}
    Value(const char *s) : kind(String), str(s) {}
    static Value array() { Value v; v.kind = Array; return v; }
    static Value object() { Value v; v.kind = Object; return v; }
    Value &operator[](const std::string &k) { kind = Object; return obj[k]; }
    void push(const Value &v) { kind = Array; arr.push_back(v); }

    void serialize(std::string &out) const
    {
        switch (kind) {
        case Null: out += "null"; break;
        case Number: {
            char buf[64];
            double ip;
            if (std::isnan(num) || std::isinf(num)) { out += "null"; break; }
            snprintf(buf, sizeof buf, (std::fabs(num) < 9007199254740992.0 && std::modf(num, &ip) == 0) ? "%.f" : "%.17g", num);
            out += buf;
            break;
        }
        case String:
            out += '"';
            for (char c : str) {
                if (c == '"' || c == '\\') { out += '\\'; out += c; }
                else if (c == '\n') out += "\\n";
                else out += c;
            }
            out += '"';
            break;
        case Array: {
            out += '[';
            for (size_t i = 0; i < arr.size(); i++) { if (i) out += ','; arr[i].serialize(out); }
            out += ']';
            break;
        }
        case Object: {
            out += '{';
            bool first = true;
            for (const auto &kv : obj) {
                if (!first) out += ',';
                first = false;
                Value(kv.first).serialize(out);
                out += ':';
                kv.second.serialize(out);
            }
            out += '}';
            break;
        }
        }
    }
    std::string serialize() const { std::string s; serialize(s); return s; }
};

} // namespace frogjson
