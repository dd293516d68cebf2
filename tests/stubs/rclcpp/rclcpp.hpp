// Test-only stand-in so the plugin shim can be syntax-checked without ROS 2.
#pragma once
#include <cstdio>
#include <string>
namespace rclcpp
{
struct Logger {};
class Node
{
public:
  template<typename T>
  T declare_parameter(const std::string &, const T & default_value) { return default_value; }
  Logger get_logger() const { return Logger(); }
};
}  // namespace rclcpp
#define RCLCPP_ERROR(logger, ...) do { (void)(logger); std::fprintf(stderr, __VA_ARGS__); } while (0)
