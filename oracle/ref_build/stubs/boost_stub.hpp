// STUB of the few Boost facilities the reference's mesh_map / planners use (boost is absent in this
// image).  Written for oracle/ref_build only: test infrastructure, never linked into the product.
//   boost::optional<T>, boost::optional<T&>, boost::none   (subset)
//   boost::uuids::random_generator / uuid / to_string      (deterministic dummy)
//   boost::adjacency_list<vecS, vecS, bidirectionalS>, add_vertex/add_edge/adjacent_vertices/in_edges/
//   source/num_vertices/topological_sort                    (what layer_manager.cpp calls)
#pragma once
#include <cassert>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include <type_traits>

namespace boost
{
struct none_t { };
static const none_t none{};

template <typename T>
class optional
{
public:
  optional() : has_(false) { }
  optional(none_t) : has_(false) { }
  optional(const T& v) : has_(true), val_(v) { }
  optional(T&& v) : has_(true), val_(std::move(v)) { }
  template <typename U, typename = std::enable_if_t<std::is_convertible<U, T>::value && !std::is_same<std::decay_t<U>, optional>::value && !std::is_same<std::decay_t<U>, none_t>::value && !std::is_same<std::decay_t<U>, T>::value>>
  optional(U&& u) : has_(true), val_(std::forward<U>(u)) { }
  explicit operator bool() const { return has_; }
  bool operator!() const { return !has_; }
  T& get() { assert(has_); return val_; }
  const T& get() const { assert(has_); return val_; }
  T& operator*() { return get(); }
  const T& operator*() const { return get(); }
  T* operator->() { return &get(); }
  const T* operator->() const { return &get(); }
  T& value() { if (!has_) throw std::runtime_error("bad optional access"); return val_; }
  const T& value() const { if (!has_) throw std::runtime_error("bad optional access"); return val_; }
  template <typename U> T value_or(U&& d) const { return has_ ? val_ : static_cast<T>(std::forward<U>(d)); }
  bool is_initialized() const { return has_; }
private:
  bool has_;
  T val_{};
};

// optional reference: rebinding semantics like boost's
template <typename T>
class optional<T&>
{
public:
  optional() : p_(nullptr) { }
  optional(none_t) : p_(nullptr) { }
  optional(T& r) : p_(&r) { }
  template <typename U, typename = std::enable_if_t<std::is_base_of<T, U>::value && !std::is_same<T, U>::value>>
  optional(U& r) : p_(&r) { }
  explicit operator bool() const { return p_ != nullptr; }
  bool operator!() const { return p_ == nullptr; }
  T& get() const { assert(p_); return *p_; }
  T& operator*() const { return get(); }
  T* operator->() const { return p_; }
  T& value() const { if (!p_) throw std::runtime_error("bad optional access"); return *p_; }
  // boost::optional<T&>::value_or takes an lvalue and returns a reference
  T& value_or(T& d) const { return p_ ? *p_ : d; }
  bool is_initialized() const { return p_ != nullptr; }
private:
  T* p_;
};

namespace uuids
{
struct uuid { uint64_t hi = 0, lo = 0; };
struct random_generator { uuid operator()() { static uint64_t n = 0; uuid u; u.lo = ++n; return u; } };
inline std::string to_string(const uuid& u) { return "stub-uuid-" + std::to_string(u.lo); }
}  // namespace uuids

// ---- graph (only the bidirectional vecS/vecS adjacency list) ----
struct vecS { };
struct bidirectionalS { };

template <typename OutS, typename VertS, typename Dir>
class adjacency_list
{
public:
  using vertex_descriptor = std::size_t;
  struct edge_descriptor { std::size_t src, dst; };
  std::vector<std::vector<std::size_t>> out, in;
};
template <typename G> struct graph_traits
{
  using vertex_descriptor = typename G::vertex_descriptor;
  using edge_descriptor = typename G::edge_descriptor;
};

template <typename G> typename G::vertex_descriptor add_vertex(G& g)
{
  g.out.emplace_back(); g.in.emplace_back();
  return g.out.size() - 1;
}
template <typename G> std::pair<typename G::edge_descriptor, bool> add_edge(std::size_t u, std::size_t v, G& g)
{
  g.out[u].push_back(v); g.in[v].push_back(u);
  return { typename G::edge_descriptor{ u, v }, true };
}
template <typename G> std::size_t num_vertices(const G& g) { return g.out.size(); }

struct adjacency_iter_stub
{
  const std::size_t* p;
  std::size_t dereference() const { return *p; }
  std::size_t operator*() const { return *p; }
  adjacency_iter_stub& operator++() { ++p; return *this; }
  adjacency_iter_stub operator++(int) { adjacency_iter_stub t = *this; ++p; return t; }
  bool operator==(const adjacency_iter_stub& o) const { return p == o.p; }
  bool operator!=(const adjacency_iter_stub& o) const { return p != o.p; }
};
template <typename G> std::pair<adjacency_iter_stub, adjacency_iter_stub> adjacent_vertices(std::size_t v, const G& g)
{
  const auto& o = g.out[v];
  return { adjacency_iter_stub{ o.data() }, adjacency_iter_stub{ o.data() + o.size() } };
}
template <typename G> struct in_edge_iter_stub
{
  std::size_t dst; const std::size_t* p;
  typename G::edge_descriptor operator*() const { return typename G::edge_descriptor{ *p, dst }; }
  in_edge_iter_stub& operator++() { ++p; return *this; }
  in_edge_iter_stub operator++(int) { in_edge_iter_stub t = *this; ++p; return t; }
  bool operator==(const in_edge_iter_stub& o) const { return p == o.p; }
  bool operator!=(const in_edge_iter_stub& o) const { return p != o.p; }
};
template <typename G> std::pair<in_edge_iter_stub<G>, in_edge_iter_stub<G>> in_edges(std::size_t v, const G& g)
{
  const auto& i = g.in[v];
  return { in_edge_iter_stub<G>{ v, i.data() }, in_edge_iter_stub<G>{ v, i.data() + i.size() } };
}
template <typename E, typename G> std::size_t source(const E& e, const G&) { return e.src; }
template <typename E, typename G> std::size_t target(const E& e, const G&) { return e.dst; }

// boost::topological_sort writes the vertices in REVERSE topological order (DFS finish order): with an
// edge u -> v ("u depends on v"), v is written before u.  Vertices are visited in index order and
// out-edges in insertion order, like boost's depth_first_search on a vecS graph.
template <typename G, typename OutIt> void topological_sort(const G& g, OutIt out)
{
  const std::size_t n = g.out.size();
  std::vector<int> color(n, 0);
  std::vector<std::pair<std::size_t, std::size_t>> stack;
  for (std::size_t s = 0; s < n; ++s) {
    if (color[s]) continue;
    color[s] = 1; stack.push_back({ s, 0 });
    while (!stack.empty()) {
      auto& [u, i] = stack.back();
      if (i < g.out[u].size()) {
        const std::size_t v = g.out[u][i++];
        if (color[v] == 0) { color[v] = 1; stack.push_back({ v, 0 }); }
        else if (color[v] == 1) throw std::runtime_error("The graph must be a DAG.");
      } else { color[u] = 2; *out++ = u; stack.pop_back(); }
    }
  }
}
}  // namespace boost
