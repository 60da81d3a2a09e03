// json.hpp -- just enough JSON (objects, arrays, strings, numbers, true/false/null) to read the reference's config
// files; the reference uses nlohmann_json, which is not installed here.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace cart::json {
struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Value> arr;
    std::map<std::string, Value> obj;

    bool is_object() const { return kind == Object; }
    bool is_array() const { return kind == Array; }
    bool contains(const std::string &k) const { return kind == Object && obj.count(k) != 0; }
    const Value &at(const std::string &k) const {
        auto it = obj.find(k);
        if (kind != Object || it == obj.end()) throw std::runtime_error("Key " + k + " not found.");
        return it->second;
    }
    template <typename T> T get() const;
};
template <> inline int Value::get<int>() const {
    if (kind != Number) throw std::runtime_error("JSON value is not a number");
    if (!(num >= -2147483648.0 && num <= 2147483647.0)) throw std::runtime_error("JSON number does not fit an int");   // also refuses NaN (the cast would be undefined)
    return (int)num;
}
template <> inline double Value::get<double>() const { if (kind != Number) throw std::runtime_error("JSON value is not a number"); return num; }
template <> inline bool Value::get<bool>() const { if (kind != Bool) throw std::runtime_error("JSON value is not a boolean"); return b; }
template <> inline std::string Value::get<std::string>() const { if (kind != String) throw std::runtime_error("JSON value is not a string"); return str; }

Value parse(const std::string &text);
}  // namespace cart::json
