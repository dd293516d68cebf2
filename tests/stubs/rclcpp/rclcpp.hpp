// Test-only stand-in so the plugin shim can be syntax-checked -- and, in
// tests/cpp/shim_runtime.cpp, exercised -- without ROS 2: a Node whose parameters are
// whatever the test put into `overrides` (the node's YAML, as strings), else the default.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
namespace rclcpp
{
struct Logger {};
class Node
{
public:
  std::map<std::string, std::string> overrides;
  std::vector<std::string> declared;

  template<typename T>
  T declare_parameter(const std::string & name, const T & default_value)
  {
    // (rclcpp throws when a parameter is declared twice: so does this)
    for (const auto & d : declared) if (d == name) throw std::runtime_error("parameter declared twice: " + name);
    declared.push_back(name);
    const auto it = overrides.find(name);
    return it == overrides.end() ? default_value : parse(it->second, static_cast<const T *>(nullptr));
  }
  Logger get_logger() const { return Logger(); }

private:
  static double parse(const std::string & s, const double *) { return std::strtod(s.c_str(), nullptr); }
  static int parse(const std::string & s, const int *) { return std::atoi(s.c_str()); }
  static bool parse(const std::string & s, const bool *) { return s == "true" || s == "1"; }
  static std::string parse(const std::string & s, const std::string *) { return s; }
  static std::vector<int64_t> parse(const std::string & s, const std::vector<int64_t> *)
  {
    std::vector<int64_t> out;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ',')) if (!tok.empty()) out.push_back(std::atoll(tok.c_str()));
    return out;
  }
};
}  // namespace rclcpp
#define RCLCPP_ERROR(logger, ...) do { (void)(logger); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
