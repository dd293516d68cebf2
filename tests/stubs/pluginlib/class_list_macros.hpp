// Test-only stand-in: registration is a no-op outside a ROS 2 workspace.
#pragma once
#define PLUGINLIB_EXPORT_CLASS(class_type, base_class_type)
