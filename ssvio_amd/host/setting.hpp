// ssvio_amd/host/setting.hpp -- the configuration store of the headless runner: ssvio::Setting
// (/root/reference/include/ssvio/setting.hpp:21-61) reads config/kitti_00.yaml through cv::FileStorage; the file is
// a flat "%YAML:1.0" map of `key: value` lines (dotted keys, numbers, quoted strings, `#` comments), which is all this
// reader accepts.  Get<T>(key) converts like cv::FileNode does: integers from reals by rounding half to even, a
// missing key yields T() -- the reference relies on that silently, so Has() is there for callers that care.
#pragma once
#include <cmath>
#include <fstream>
#include <stdexcept>
#include <string>
#include <unordered_map>

namespace ssx::host {

class Setting {
 public:
  Setting() = default;
  explicit Setting(const std::string& path) { Load(path); }

  void Load(const std::string& path)
  {
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("Setting: cannot open " + path);
    std::string line;
    while (std::getline(f, line)) ParseLine(line);
  }
  void ParseLine(std::string line)
  {
    // strip a comment that is not inside quotes
    bool quoted = false;
    for (size_t i = 0; i < line.size(); ++i) {
      if (line[i] == '"') quoted = !quoted;
      if (line[i] == '#' && !quoted) { line.resize(i); break; }
    }
    const size_t colon = line.find(':');
    if (line.empty() || line[0] == '%' || line.rfind("---", 0) == 0 || colon == std::string::npos) return;
    std::string key = Trim(line.substr(0, colon)), val = Trim(line.substr(colon + 1));
    if (key.empty()) return;
    if (val.size() >= 2 && val.front() == '"' && val.back() == '"') val = val.substr(1, val.size() - 2);
    values_[key] = val;
  }
  void Set(const std::string& key, const std::string& value) { values_[key] = value; }
  bool Has(const std::string& key) const { return values_.count(key) != 0; }

  template <typename T> T Get(const std::string& key) const;

 private:
  static std::string Trim(const std::string& s)
  {
    const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
  }
  double Number(const std::string& key) const
  {
    auto it = values_.find(key);
    if (it == values_.end() || it->second.empty()) return 0.0;
    try { return std::stod(it->second); } catch (...) { throw std::runtime_error("Setting: " + key + " is not a number: " + it->second); }
  }
  std::unordered_map<std::string, std::string> values_;
};

template <> inline double Setting::Get<double>(const std::string& key) const { return Number(key); }
template <> inline float Setting::Get<float>(const std::string& key) const { return (float)Number(key); }
template <> inline int Setting::Get<int>(const std::string& key) const { return (int)std::nearbyint(Number(key)); }
template <> inline std::string Setting::Get<std::string>(const std::string& key) const
{
  auto it = values_.find(key);
  return it == values_.end() ? std::string() : it->second;
}

}  // namespace ssx::host
