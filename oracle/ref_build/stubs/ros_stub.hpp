// STUB of the ROS 2 / MBF / tf2 / pluginlib surface the reference's planners, mesh_map and mesh_layers
// touch (none of it exists in this image).  Everything here is plumbing: parameters come from a table the
// harness fills, publishers drop their messages, the tf buffer is the identity.  The only arithmetic is
// tf2's Matrix3x3 <-> Quaternion conversion (published Bullet/tf2 formulas, doubles), which
// mesh_map::calculatePoseFromDirection uses.  Test infrastructure for oracle/ref_build only.
#pragma once
// tf2/LinearMath/Scalar.h (pulled in by mesh_map/util.h -> tf2/LinearMath/Vector3.h and by tf2_ros/buffer.h)
// includes <math.h>.  With libstdc++ that header injects the float/long double overloads of sqrt, acos, exp,
// cos ... into the GLOBAL namespace, so the reference's unqualified `sqrt(float)` calls (inflation_layer.cpp:
// 189,203,206) run in float exactly as in the real build; without it they would silently promote to double.
#include <math.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <functional>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <shared_mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_set>
#include <variant>
#include <vector>

// ======================================================================================================
// messages
// ======================================================================================================
namespace builtin_interfaces::msg { struct Time { int32_t sec = 0; uint32_t nanosec = 0; }; }

namespace rclcpp
{
class Time
{
public:
  Time() : ns_(0) { }
  explicit Time(int64_t ns) : ns_(ns) { }
  Time(const builtin_interfaces::msg::Time& t) : ns_(int64_t(t.sec) * 1000000000ll + t.nanosec) { }
  int64_t nanoseconds() const { return ns_; }
  double seconds() const { return ns_ * 1e-9; }
  operator builtin_interfaces::msg::Time() const
  {
    builtin_interfaces::msg::Time t; t.sec = int32_t(ns_ / 1000000000ll); t.nanosec = uint32_t(ns_ % 1000000000ll); return t;
  }
private:
  int64_t ns_;
};
class Duration
{
public:
  explicit Duration(int64_t ns = 0) : ns_(ns) { }
  static Duration from_seconds(double s) { return Duration(int64_t(s * 1e9)); }
  int64_t nanoseconds() const { return ns_; }
private:
  int64_t ns_;
};
}  // namespace rclcpp

namespace std_msgs::msg
{
struct Header { builtin_interfaces::msg::Time stamp; std::string frame_id; };
struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; };
}
namespace geometry_msgs::msg
{
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::msg::Header header; Pose pose; };
struct PointStamped { std_msgs::msg::Header header; Point point; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::msg::Header header; std::string child_frame_id; Transform transform; };
}
namespace nav_msgs::msg { struct Path { std_msgs::msg::Header header; std::vector<geometry_msgs::msg::PoseStamped> poses; }; }
namespace visualization_msgs::msg
{
struct Marker
{
  enum : int32_t { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, LINE_STRIP = 4, LINE_LIST = 5, TEXT_VIEW_FACING = 9, TRIANGLE_LIST = 11 };
  enum : int32_t { ADD = 0, MODIFY = 0, DELETE = 2 };
  std_msgs::msg::Header header; std::string ns; int32_t id = 0; int32_t type = 0; int32_t action = 0;
  geometry_msgs::msg::Pose pose; geometry_msgs::msg::Vector3 scale; std_msgs::msg::ColorRGBA color;
  std::vector<geometry_msgs::msg::Point> points; std::vector<std_msgs::msg::ColorRGBA> colors; std::string text;
};
struct MarkerArray { std::vector<Marker> markers; };
}
namespace mesh_msgs::msg
{
struct MeshGeometry { };
struct MeshGeometryStamped { std_msgs::msg::Header header; std::string uuid; MeshGeometry mesh_geometry; };
struct MeshVertexColors { std::vector<std_msgs::msg::ColorRGBA> vertex_colors; };
struct MeshVertexColorsStamped { std_msgs::msg::Header header; std::string uuid; MeshVertexColors mesh_vertex_colors; };
struct MeshVertexCosts { std::vector<float> costs; };
struct MeshVertexCostsStamped { std_msgs::msg::Header header; std::string uuid; std::string type; MeshVertexCosts mesh_vertex_costs; };
struct MeshVertexCostsSparseStamped { std_msgs::msg::Header header; std::string uuid; std::string type; };
}
namespace std_srvs::srv { struct Trigger { struct Request { }; struct Response { bool success = false; std::string message; }; }; }
namespace mbf_msgs::action
{
struct GetPath
{
  struct Result
  {
    enum : uint32_t { SUCCESS = 0, FAILURE = 50, CANCELED = 51, INVALID_START = 52, INVALID_GOAL = 53, NO_PATH_FOUND = 54,
                      PAT_EXCEEDED = 55, EMPTY_PATH = 56, TF_ERROR = 57, NOT_INITIALIZED = 58, INVALID_PLUGIN = 59,
                      INTERNAL_ERROR = 60 };
  };
};
}
namespace rcl_interfaces::msg
{
struct FloatingPointRange { double from_value = 0, to_value = 0, step = 0; };
struct IntegerRange { int64_t from_value = 0, to_value = 0; uint64_t step = 0; };
struct ParameterType { enum : uint8_t { PARAMETER_NOT_SET = 0, PARAMETER_BOOL = 1, PARAMETER_INTEGER = 2, PARAMETER_DOUBLE = 3, PARAMETER_STRING = 4, PARAMETER_STRING_ARRAY = 9 }; };
struct ParameterDescriptor
{
  std::string name; uint8_t type = 0; std::string description;
  std::vector<FloatingPointRange> floating_point_range; std::vector<IntegerRange> integer_range;
};
struct SetParametersResult { bool successful = false; std::string reason; };
}

// ======================================================================================================
// rclcpp
// ======================================================================================================
namespace rclcpp
{
enum ParameterType : uint8_t { PARAMETER_NOT_SET = 0, PARAMETER_BOOL = 1, PARAMETER_INTEGER = 2, PARAMETER_DOUBLE = 3, PARAMETER_STRING = 4, PARAMETER_STRING_ARRAY = 9 };

namespace exceptions
{
class InvalidParametersException : public std::runtime_error { public: using std::runtime_error::runtime_error; };
class ParameterUninitializedException : public std::runtime_error { public: using std::runtime_error::runtime_error; };
}

using ParameterValue = std::variant<std::monostate, bool, int64_t, double, std::string, std::vector<std::string>>;

class Parameter
{
public:
  Parameter() { }
  Parameter(std::string name, ParameterValue v) : name_(std::move(name)), v_(std::move(v)) { }
  const std::string& get_name() const { return name_; }
  bool as_bool() const { return std::get<bool>(v_); }
  int64_t as_int() const { return std::get<int64_t>(v_); }
  double as_double() const { if (auto* i = std::get_if<int64_t>(&v_)) return double(*i); return std::get<double>(v_); }
  std::string as_string() const { return std::get<std::string>(v_); }
  std::vector<std::string> as_string_array() const
  {
    if (std::holds_alternative<std::monostate>(v_)) return { };
    return std::get<std::vector<std::string>>(v_);
  }
private:
  std::string name_;
  ParameterValue v_;
};

class Logger
{
public:
  explicit Logger(std::string n = "") : name_(std::move(n)) { }
  Logger get_child(const std::string& s) const { return Logger(name_ + "." + s); }
  const std::string& name() const { return name_; }
private:
  std::string name_;
};
inline Logger get_logger(const std::string& n) { return Logger(n); }

// log sink: silent unless REF_STUB_LOG is set; errors always counted
struct LogState { static int& verbosity() { static int v = -1; return v; } static std::atomic<long>& errors() { static std::atomic<long> e{ 0 }; return e; } };
inline void stub_log(int level, const Logger& l, const std::string& msg)
{
  int& v = LogState::verbosity();
  if (v < 0) { const char* e = std::getenv("REF_STUB_LOG"); v = e ? std::atoi(e) : 0; }
  if (level >= 3) LogState::errors()++;
  if (v > 0 && level >= 4 - v) std::fprintf(stderr, "[ref %d] %s: %s\n", level, l.name().c_str(), msg.c_str());
}
inline std::string stub_format(const char* fmt, ...)
{
  char buf[2048];
  va_list ap; va_start(ap, fmt); std::vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  return buf;
}

class Clock
{
public:
  Time now() const
  {
    const auto t = std::chrono::system_clock::now().time_since_epoch();
    return Time(std::chrono::duration_cast<std::chrono::nanoseconds>(t).count());
  }
};

class QoS
{
public:
  QoS(std::size_t depth = 1) { (void)depth; }
  QoS& transient_local() { return *this; }
};

// STUB publishers RECORD: the last message and a count, so that a test can look at what a planner published.  Several
// publishers may exist for one topic (the reference planner's and a plugin's `~/path`): a look-up returns the one that
// published last.
struct PublisherBase
{
  virtual ~PublisherBase() = default;
  std::string topic_;
  std::size_t count_ = 0;
  unsigned long long seq_ = 0;          // global sequence number of its last publish()
};
inline std::vector<PublisherBase*>& stub_publishers() { static std::vector<PublisherBase*> v; return v; }
inline unsigned long long& stub_publish_seq() { static unsigned long long s = 0; return s; }
inline PublisherBase* stub_find_publisher(const std::string& topic)
{
  PublisherBase* best = nullptr;
  for (PublisherBase* p : stub_publishers()) if (p->topic_ == topic && (!best || p->seq_ >= best->seq_)) best = p;
  return best;
}
template <typename MsgT>
class Publisher : public PublisherBase
{
public:
  using SharedPtr = std::shared_ptr<Publisher<MsgT>>;
  explicit Publisher(std::string topic) { topic_ = std::move(topic); stub_publishers().push_back(this); }
  ~Publisher() override { auto& v = stub_publishers(); v.erase(std::remove(v.begin(), v.end(), static_cast<PublisherBase*>(this)), v.end()); }
  void publish(const MsgT& msg) { ++count_; seq_ = ++stub_publish_seq(); last_ = msg; }
  std::size_t get_subscription_count() const { return 0; }
  std::size_t get_intra_process_subscription_count() const { return 0; }
  const char* get_topic_name() const { return topic_.c_str(); }
  const MsgT& last() const { return last_; }
private:
  MsgT last_{};
};
template <typename SrvT> class Service { public: using SharedPtr = std::shared_ptr<Service<SrvT>>; };
class TimerBase { public: using SharedPtr = std::shared_ptr<TimerBase>; };

namespace node_interfaces
{
struct OnSetParametersCallbackHandle
{
  using SharedPtr = std::shared_ptr<OnSetParametersCallbackHandle>;
  std::function<rcl_interfaces::msg::SetParametersResult(const std::vector<Parameter>&)> callback;
};
}

class Node
{
public:
  using SharedPtr = std::shared_ptr<Node>;
  explicit Node(std::string name = "ref_node") : logger_(std::move(name)), clock_(std::make_shared<Clock>()) { }

  // ---- parameter table (filled by the harness before the reference code declares them) ----
  void stub_set_override(const std::string& name, ParameterValue v) { overrides_[name] = std::move(v); }
  // set a declared parameter and fire the on-set callbacks, like `ros2 param set`
  bool stub_set_parameter(const std::string& name, ParameterValue v)
  {
    values_[name] = v;
    const std::vector<Parameter> ps{ Parameter(name, v) };
    bool ok = true;
    const auto cbs = callbacks_;   // callbacks may register further callbacks
    for (const auto& w : cbs) if (auto h = w.lock()) ok = h->callback(ps).successful && ok;
    return ok;
  }

  template <typename T> static ParameterValue to_value(const T& v)
  {
    if constexpr (std::is_same<T, bool>::value) return ParameterValue(v);
    else if constexpr (std::is_integral<T>::value) return ParameterValue(int64_t(v));
    else if constexpr (std::is_floating_point<T>::value) return ParameterValue(double(v));
    else if constexpr (std::is_same<T, std::vector<std::string>>::value) return ParameterValue(v);
    else return ParameterValue(std::string(v));
  }
  template <typename T> static T from_value(const ParameterValue& v)
  {
    if constexpr (std::is_same<T, bool>::value) return std::get<bool>(v);
    else if constexpr (std::is_integral<T>::value) return T(std::get<int64_t>(v));
    else if constexpr (std::is_floating_point<T>::value) { if (auto* i = std::get_if<int64_t>(&v)) return T(*i); return T(std::get<double>(v)); }
    else if constexpr (std::is_same<T, std::vector<std::string>>::value) return std::get<std::vector<std::string>>(v);
    else return std::get<std::string>(v);
  }

  // declare_parameter(name, default[, descriptor]) -> value (override wins)
  template <typename T>
  auto declare_parameter(const std::string& name, const T& default_value,
                         const rcl_interfaces::msg::ParameterDescriptor& = rcl_interfaces::msg::ParameterDescriptor())
  {
    using V = std::conditional_t<std::is_convertible<T, std::string>::value && !std::is_arithmetic<T>::value, std::string, T>;
    auto it = overrides_.find(name);
    if (it != overrides_.end()) { values_[name] = it->second; return from_value<V>(it->second); }
    values_[name] = to_value<V>(V(default_value));
    return V(default_value);
  }
  // declare_parameter<T>(name[, descriptor]) without default: must be overridden
  template <typename T>
  T declare_parameter(const std::string& name,
                      const rcl_interfaces::msg::ParameterDescriptor& = rcl_interfaces::msg::ParameterDescriptor())
  {
    auto it = overrides_.find(name);
    if (it == overrides_.end()) throw exceptions::ParameterUninitializedException("parameter '" + name + "' is not set");
    values_[name] = it->second;
    return from_value<T>(it->second);
  }
  template <typename T> bool get_parameter(const std::string& name, T& out) const
  {
    auto it = values_.find(name);
    if (it == values_.end()) { it = overrides_.find(name); if (it == overrides_.end()) return false; }
    out = from_value<T>(it->second);
    return true;
  }
  Parameter get_parameter(const std::string& name) const
  {
    auto it = values_.find(name);
    if (it == values_.end()) { it = overrides_.find(name); if (it == overrides_.end()) return Parameter(name, ParameterValue()); }
    return Parameter(name, it->second);
  }

  template <typename F>
  node_interfaces::OnSetParametersCallbackHandle::SharedPtr add_on_set_parameters_callback(F f)
  {
    auto h = std::make_shared<node_interfaces::OnSetParametersCallbackHandle>();
    h->callback = [f](const std::vector<Parameter>& p) { return f(p); };
    callbacks_.push_back(h);
    return h;
  }

  const Logger& get_logger() const { return logger_; }
  Time now() const { return clock_->now(); }
  std::shared_ptr<Clock> get_clock() const { return clock_; }

  template <typename MsgT> typename Publisher<MsgT>::SharedPtr create_publisher(const std::string& topic, const QoS& = QoS())
  {
    return std::make_shared<Publisher<MsgT>>(topic);
  }
  template <typename SrvT, typename F> typename Service<SrvT>::SharedPtr create_service(const std::string&, F) { return std::make_shared<Service<SrvT>>(); }
  template <typename D, typename F> TimerBase::SharedPtr create_wall_timer(D, F) { return std::make_shared<TimerBase>(); }

private:
  Logger logger_;
  std::shared_ptr<Clock> clock_;
  std::map<std::string, ParameterValue> overrides_, values_;
  std::vector<std::weak_ptr<node_interfaces::OnSetParametersCallbackHandle>> callbacks_;
};
}  // namespace rclcpp

#define REF_STUB_LOG_STREAM(level, logger, args) do { std::ostringstream ref_ss_; ref_ss_ << args; ::rclcpp::stub_log(level, logger, ref_ss_.str()); } while (0)
#define REF_STUB_LOG_FMT(level, logger, ...) ::rclcpp::stub_log(level, logger, ::rclcpp::stub_format(__VA_ARGS__))
#define RCLCPP_DEBUG_STREAM(logger, args) REF_STUB_LOG_STREAM(0, logger, args)
#define RCLCPP_INFO_STREAM(logger, args) REF_STUB_LOG_STREAM(1, logger, args)
#define RCLCPP_WARN_STREAM(logger, args) REF_STUB_LOG_STREAM(2, logger, args)
#define RCLCPP_ERROR_STREAM(logger, args) REF_STUB_LOG_STREAM(3, logger, args)
#define RCLCPP_FATAL_STREAM(logger, args) REF_STUB_LOG_STREAM(3, logger, args)
#define RCLCPP_DEBUG(logger, ...) REF_STUB_LOG_FMT(0, logger, __VA_ARGS__)
#define RCLCPP_INFO(logger, ...) REF_STUB_LOG_FMT(1, logger, __VA_ARGS__)
#define RCLCPP_WARN(logger, ...) REF_STUB_LOG_FMT(2, logger, __VA_ARGS__)
#define RCLCPP_ERROR(logger, ...) REF_STUB_LOG_FMT(3, logger, __VA_ARGS__)
#define RCLCPP_FATAL(logger, ...) REF_STUB_LOG_FMT(3, logger, __VA_ARGS__)
#define RCLCPP_ERROR_THROTTLE(logger, clock, period, ...) do { (void)(clock); REF_STUB_LOG_FMT(3, logger, __VA_ARGS__); } while (0)
#define RCLCPP_INFO_SKIPFIRST_THROTTLE(logger, clock, period, ...) do { (void)(clock); REF_STUB_LOG_FMT(1, logger, __VA_ARGS__); } while (0)
#define RCLCPP_DEBUG_STREAM_THROTTLE(logger, clock, period, args) do { (void)(clock); REF_STUB_LOG_STREAM(0, logger, args); } while (0)
#define RCLCPP_ERROR_STREAM_THROTTLE(logger, clock, period, args) do { (void)(clock); REF_STUB_LOG_STREAM(3, logger, args); } while (0)

// ======================================================================================================
// tf2 (LinearMath subset with the published Bullet formulas), tf2_ros::Buffer (identity), conversions
// ======================================================================================================
namespace tf2
{
using tf2Scalar = double;
class TransformException : public std::runtime_error { public: using std::runtime_error::runtime_error; };

class Vector3
{
public:
  Vector3() : v_{ 0, 0, 0 } { }
  Vector3(double x, double y, double z) : v_{ x, y, z } { }
  double x() const { return v_[0]; } double y() const { return v_[1]; } double z() const { return v_[2]; }
  double operator[](int i) const { return v_[i]; }
  double& operator[](int i) { return v_[i]; }
  double dot(const Vector3& o) const { return v_[0] * o.v_[0] + v_[1] * o.v_[1] + v_[2] * o.v_[2]; }
private:
  double v_[3];
};

class Quaternion
{
public:
  Quaternion() : q_{ 0, 0, 0, 1 } { }
  Quaternion(double x, double y, double z, double w) : q_{ x, y, z, w } { }
  double x() const { return q_[0]; } double y() const { return q_[1]; } double z() const { return q_[2]; } double w() const { return q_[3]; }
  void setValue(double x, double y, double z, double w) { q_[0] = x; q_[1] = y; q_[2] = z; q_[3] = w; }
  double length2() const { return q_[0] * q_[0] + q_[1] * q_[1] + q_[2] * q_[2] + q_[3] * q_[3]; }
  double length() const { return std::sqrt(length2()); }
  Quaternion& operator/=(double s) { const double inv = 1.0 / s; q_[0] *= inv; q_[1] *= inv; q_[2] *= inv; q_[3] *= inv; return *this; }
  Quaternion& normalize() { return *this /= length(); }
private:
  double q_[4];
};

class Matrix3x3
{
public:
  Matrix3x3() { setIdentity(); }
  Matrix3x3(double xx, double xy, double xz, double yx, double yy, double yz, double zx, double zy, double zz) { setValue(xx, xy, xz, yx, yy, yz, zx, zy, zz); }
  void setIdentity() { setValue(1, 0, 0, 0, 1, 0, 0, 0, 1); }
  void setValue(double xx, double xy, double xz, double yx, double yy, double yz, double zx, double zy, double zz)
  {
    m_[0] = Vector3(xx, xy, xz); m_[1] = Vector3(yx, yy, yz); m_[2] = Vector3(zx, zy, zz);
  }
  const Vector3& operator[](int i) const { return m_[i]; }
  void setRotation(const Quaternion& q)
  {
    const double d = q.length2();
    const double s = 2.0 / d;
    const double xs = q.x() * s, ys = q.y() * s, zs = q.z() * s;
    const double wx = q.w() * xs, wy = q.w() * ys, wz = q.w() * zs;
    const double xx = q.x() * xs, xy = q.x() * ys, xz = q.x() * zs;
    const double yy = q.y() * ys, yz = q.y() * zs, zz = q.z() * zs;
    setValue(1.0 - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0 - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0 - (xx + yy));
  }
  void getRotation(Quaternion& q) const
  {
    const double trace = m_[0].x() + m_[1].y() + m_[2].z();
    double temp[4];
    if (trace > 0.0) {
      double s = std::sqrt(trace + 1.0);
      temp[3] = s * 0.5;
      s = 0.5 / s;
      temp[0] = (m_[2].y() - m_[1].z()) * s;
      temp[1] = (m_[0].z() - m_[2].x()) * s;
      temp[2] = (m_[1].x() - m_[0].y()) * s;
    } else {
      const int i = m_[0].x() < m_[1].y() ? (m_[1].y() < m_[2].z() ? 2 : 1) : (m_[0].x() < m_[2].z() ? 2 : 0);
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      double s = std::sqrt(m_[i][i] - m_[j][j] - m_[k][k] + 1.0);
      temp[i] = s * 0.5;
      s = 0.5 / s;
      temp[3] = (m_[k][j] - m_[j][k]) * s;
      temp[j] = (m_[j][i] + m_[i][j]) * s;
      temp[k] = (m_[k][i] + m_[i][k]) * s;
    }
    q.setValue(temp[0], temp[1], temp[2], temp[3]);
  }
  Vector3 operator*(const Vector3& v) const { return Vector3(m_[0].dot(v), m_[1].dot(v), m_[2].dot(v)); }
private:
  Vector3 m_[3];
};

class Transform
{
public:
  Transform() { }
  void setBasis(const Matrix3x3& b) { basis_ = b; }
  const Matrix3x3& getBasis() const { return basis_; }
  void setOrigin(const Vector3& o) { origin_ = o; }
  const Vector3& getOrigin() const { return origin_; }
  Quaternion getRotation() const { Quaternion q; basis_.getRotation(q); return q; }
  void setRotation(const Quaternion& q) { basis_.setRotation(q); }
private:
  Matrix3x3 basis_;
  Vector3 origin_;
};

inline geometry_msgs::msg::Pose& toMsg(const Transform& in, geometry_msgs::msg::Pose& out)
{
  out.position.x = in.getOrigin().x(); out.position.y = in.getOrigin().y(); out.position.z = in.getOrigin().z();
  const Quaternion q = in.getRotation();
  out.orientation.x = q.x(); out.orientation.y = q.y(); out.orientation.z = q.z(); out.orientation.w = q.w();
  return out;
}
inline void fromMsg(const geometry_msgs::msg::Transform& in, Transform& out)
{
  out.setOrigin(Vector3(in.translation.x, in.translation.y, in.translation.z));
  out.setRotation(Quaternion(in.rotation.x, in.rotation.y, in.rotation.z, in.rotation.w));
}
// identity-only transform (the harness always plans in the map frame)
inline void doTransform(const geometry_msgs::msg::PoseStamped& in, geometry_msgs::msg::PoseStamped& out, const geometry_msgs::msg::TransformStamped& t)
{
  out = in; out.header.frame_id = t.header.frame_id;
}
}  // namespace tf2

namespace tf2_ros
{
class Buffer
{
public:
  geometry_msgs::msg::TransformStamped lookupTransform(const std::string& target, const std::string& source, const rclcpp::Time&, const rclcpp::Duration&) const
  {
    if (fail_lookups) throw tf2::TransformException("stub: no transform from '" + source + "' to '" + target + "'");
    geometry_msgs::msg::TransformStamped t; t.header.frame_id = target; t.child_frame_id = source; return t;
  }
  bool fail_lookups = false;
};
}  // namespace tf2_ros

// ======================================================================================================
// pluginlib: a process-local registry filled by PLUGINLIB_EXPORT_CLASS
// ======================================================================================================
namespace pluginlib
{
class LibraryLoadException : public std::runtime_error { public: using std::runtime_error::runtime_error; };

template <typename BaseT>
struct Registry
{
  static std::map<std::string, std::function<std::shared_ptr<BaseT>()>>& get()
  {
    static std::map<std::string, std::function<std::shared_ptr<BaseT>()>> r;
    return r;
  }
};
template <typename BaseT, typename ClassT>
struct Registrar
{
  explicit Registrar(const char* name) { Registry<BaseT>::get()[name] = [] { return std::shared_ptr<BaseT>(new ClassT()); }; }
};

template <typename BaseT>
class ClassLoader
{
public:
  ClassLoader(const std::string&, const std::string&) { }
  std::shared_ptr<BaseT> createSharedInstance(const std::string& lookup_name)
  {
    // lookup names in the plugin XML are "pkg/Class"; the class is registered as "pkg::Class"
    std::string key = lookup_name;
    for (std::size_t p; (p = key.find('/')) != std::string::npos;) key.replace(p, 1, "::");
    auto& r = Registry<BaseT>::get();
    auto it = r.find(key);
    if (it == r.end()) throw LibraryLoadException("stub pluginlib: no class '" + lookup_name + "'");
    return it->second();
  }
};
}  // namespace pluginlib
#define REF_STUB_CAT2(a, b) a##b
#define REF_STUB_CAT(a, b) REF_STUB_CAT2(a, b)
#define PLUGINLIB_EXPORT_CLASS(cls, base) \
  namespace { static ::pluginlib::Registrar<base, cls> REF_STUB_CAT(ref_stub_registrar_, __COUNTER__)(#cls); }

// ======================================================================================================
// move_base_flex abstract planner base
// ======================================================================================================
namespace mbf_abstract_core
{
class AbstractPlanner
{
public:
  typedef std::shared_ptr<mbf_abstract_core::AbstractPlanner> Ptr;
  virtual ~AbstractPlanner() { }
  virtual uint32_t makePlan(const geometry_msgs::msg::PoseStamped& start, const geometry_msgs::msg::PoseStamped& goal,
                            double tolerance, std::vector<geometry_msgs::msg::PoseStamped>& plan, double& cost,
                            std::string& message) = 0;
  virtual bool cancel() = 0;
protected:
  AbstractPlanner() { }
};
}  // namespace mbf_abstract_core
