// params.hpp -- Athena-style input deck reader ("<block>" headers, "key = value" lines,
// '#' comments) with "block/key=value" command-line overrides.  Plays the role of
// Parthenon's ParameterInput for the options Hydro::Initialize reads
// (src/hydro/hydro.cpp:264-826): GetOrAdd* semantics with the same defaults, Get* throws on
// a missing key like the reference aborts.
#pragma once

#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace apk {

class ParameterInput {
 public:
  void LoadFromString(const std::string &text) {
    std::istringstream in(text);
    std::string line, block;
    while (std::getline(in, line)) {
      const auto hash = line.find('#');
      if (hash != std::string::npos) line.erase(hash);
      line = Trim(line);
      if (line.empty()) continue;
      if (line.front() == '<') {
        const auto close = line.find('>');
        if (close == std::string::npos) throw std::runtime_error("deck: unterminated block header: " + line);
        block = Trim(line.substr(1, close - 1));
        continue;
      }
      const auto eq = line.find('=');
      if (eq == std::string::npos) throw std::runtime_error("deck: expected key = value: " + line);
      if (block.empty()) throw std::runtime_error("deck: key outside of a block: " + line);
      values_[block + "/" + Trim(line.substr(0, eq))] = Trim(line.substr(eq + 1));
    }
  }
  // "block/key=value" (block may itself contain '/', e.g. parthenon/mesh/nx1=64)
  void ApplyOverride(const std::string &arg) {
    const auto eq = arg.find('=');
    if (eq == std::string::npos) throw std::runtime_error("override must be block/key=value: " + arg);
    const std::string path = Trim(arg.substr(0, eq));
    if (path.find('/') == std::string::npos) throw std::runtime_error("override must be block/key=value: " + arg);
    values_[path] = Trim(arg.substr(eq + 1));
  }
  // names of the blocks that start with `prefix` (e.g. "parthenon/output"), in deck order of
  // their names
  std::vector<std::string> BlocksWithPrefix(const std::string &prefix) const {
    std::vector<std::string> out;
    for (const auto &kv : values_) {
      const std::string block = kv.first.substr(0, kv.first.rfind('/'));
      if (block.compare(0, prefix.size(), prefix) == 0 && (out.empty() || out.back() != block)) out.push_back(block);
    }
    return out;
  }
  bool DoesParameterExist(const std::string &block, const std::string &key) const {
    return values_.count(block + "/" + key) > 0;
  }
  std::string GetString(const std::string &block, const std::string &key) const {
    auto it = values_.find(block + "/" + key);
    if (it == values_.end()) throw std::runtime_error("deck: missing required parameter <" + block + "> " + key);
    return it->second;
  }
  double GetReal(const std::string &b, const std::string &k) const { return ToReal(GetString(b, k), b, k); }
  int GetInteger(const std::string &b, const std::string &k) const { return (int)ToInt(GetString(b, k), b, k); }
  bool GetBoolean(const std::string &b, const std::string &k) const { return ToBool(GetString(b, k), b, k); }
  std::string GetOrAddString(const std::string &b, const std::string &k, const std::string &def) {
    if (!DoesParameterExist(b, k)) values_[b + "/" + k] = def;
    return GetString(b, k);
  }
  double GetOrAddReal(const std::string &b, const std::string &k, double def) {
    if (!DoesParameterExist(b, k)) {
      std::ostringstream os;
      os.precision(17);
      os << def;
      values_[b + "/" + k] = os.str();
      return def;
    }
    return GetReal(b, k);
  }
  int GetOrAddInteger(const std::string &b, const std::string &k, int def) {
    if (!DoesParameterExist(b, k)) {
      values_[b + "/" + k] = std::to_string(def);
      return def;
    }
    return GetInteger(b, k);
  }
  bool GetOrAddBoolean(const std::string &b, const std::string &k, bool def) {
    if (!DoesParameterExist(b, k)) {
      values_[b + "/" + k] = def ? "true" : "false";
      return def;
    }
    return GetBoolean(b, k);
  }

 private:
  static std::string Trim(const std::string &s) {
    const char *ws = " \t\r\n";
    const auto a = s.find_first_not_of(ws);
    if (a == std::string::npos) return "";
    const auto b = s.find_last_not_of(ws);
    return s.substr(a, b - a + 1);
  }
  static double ToReal(const std::string &v, const std::string &b, const std::string &k) {
    char *end = nullptr;
    const double x = std::strtod(v.c_str(), &end);
    if (end == v.c_str()) throw std::runtime_error("deck: <" + b + "> " + k + " is not a number: " + v);
    return x;
  }
  static long ToInt(const std::string &v, const std::string &b, const std::string &k) {
    char *end = nullptr;
    const long x = std::strtol(v.c_str(), &end, 10);
    if (end == v.c_str()) throw std::runtime_error("deck: <" + b + "> " + k + " is not an integer: " + v);
    return x;
  }
  static bool ToBool(const std::string &v, const std::string &b, const std::string &k) {
    if (v == "true" || v == "True" || v == "1") return true;
    if (v == "false" || v == "False" || v == "0") return false;
    throw std::runtime_error("deck: <" + b + "> " + k + " is not a boolean: " + v);
  }
  std::map<std::string, std::string> values_;
};

}  // namespace apk
