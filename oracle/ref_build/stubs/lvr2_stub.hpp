// STUB of the parts of lvr2 (https://github.com/uos/lvr2, un-vendored, unpinned "main") and of the
// pmp::SurfaceMesh it wraps that the reference's planners, mesh_map and mesh_layers call.  Written from
// the published lvr2 / pmp-library behaviour so that the reference's own translation units compile and run
// UNMODIFIED in this image (oracle/ref_build/build.sh).  Test infrastructure only.
//
// What is modelled faithfully (because results depend on it):
//   * pmp::SurfaceMesh::add_face / new_edge / adjust_outgoing_halfedge (the OpenMesh algorithm) -> edge ids
//     (order of first appearance, halfedge pairs), face vertex order, circulator start and CCW rotation;
//   * lvr2::Meap: array binary heap + key->index map, insert = insert-or-update, strict comparisons;
//   * lvr2 attribute maps (VectorMap with optional default value, HashMap), StableVector-like semantics;
//   * BaseVector<float> / Normal<float> float arithmetic; calcFaceNormals / calcVertexNormals /
//     calcVertexDistances as published in lvr2's NormalAlgorithms / GeometryAlgorithms.
// What is a CONVENTION here (lvr2 source not available to check): BaseVector::rotated (Rodrigues), the
// exact operation order inside normalize(), Meap tie behaviour beyond "binary heap with strict <".
#pragma once
#include <algorithm>
#include <any>
#include <array>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "boost_stub.hpp"

// ======================================================================================================
// pmp handles + SurfaceMesh connectivity
// ======================================================================================================
namespace pmp
{
using IndexType = std::uint32_t;
constexpr IndexType PMP_MAX_INDEX = std::numeric_limits<IndexType>::max();

class Handle
{
public:
  explicit Handle(IndexType idx = PMP_MAX_INDEX) : idx_(idx) { }
  IndexType idx() const { return idx_; }
  void reset() { idx_ = PMP_MAX_INDEX; }
  bool is_valid() const { return idx_ != PMP_MAX_INDEX; }
  bool operator==(const Handle& o) const { return idx_ == o.idx_; }
  bool operator!=(const Handle& o) const { return idx_ != o.idx_; }
  bool operator<(const Handle& o) const { return idx_ < o.idx_; }
protected:
  IndexType idx_;
};
class Vertex : public Handle { public: using Handle::Handle; Vertex(std::size_t i) : Handle(static_cast<IndexType>(i)) { } Vertex() : Handle() { }
  Vertex(int i) : Handle(static_cast<IndexType>(i)) { } Vertex(IndexType i) : Handle(i) { } };
class Halfedge : public Handle { public: Halfedge() : Handle() { } explicit Halfedge(IndexType i) : Handle(i) { }
  Halfedge opposite() const { return Halfedge((idx_ & 1) ? idx_ - 1 : idx_ + 1); } };
class Edge : public Handle { public: Edge() : Handle() { } explicit Edge(IndexType i) : Handle(i) { } Edge(std::size_t i) : Handle(static_cast<IndexType>(i)) { } };
class Face : public Handle { public: Face() : Handle() { } explicit Face(IndexType i) : Handle(i) { } Face(std::size_t i) : Handle(static_cast<IndexType>(i)) { } };

inline std::ostream& operator<<(std::ostream& os, Vertex v) { return os << 'v' << v.idx(); }
inline std::ostream& operator<<(std::ostream& os, Halfedge h) { return os << 'h' << h.idx(); }
inline std::ostream& operator<<(std::ostream& os, Edge e) { return os << 'e' << e.idx(); }
inline std::ostream& operator<<(std::ostream& os, Face f) { return os << 'f' << f.idx(); }

class TopologyException : public std::runtime_error { public: using std::runtime_error::runtime_error; };

class SurfaceMesh
{
public:
  // ---- element creation ----
  Vertex add_vertex_slot() { vconn_.emplace_back(); return Vertex(static_cast<IndexType>(vconn_.size() - 1)); }
  std::size_t n_vertices() const { return vconn_.size(); }
  std::size_t n_edges() const { return hconn_.size() / 2; }
  std::size_t n_halfedges() const { return hconn_.size(); }
  std::size_t n_faces() const { return fconn_.size(); }

  // ---- connectivity accessors (pmp-library names) ----
  Halfedge halfedge(Vertex v) const { return vconn_[v.idx()].halfedge_; }
  void set_halfedge(Vertex v, Halfedge h) { vconn_[v.idx()].halfedge_ = h; }
  Halfedge halfedge(Face f) const { return fconn_[f.idx()].halfedge_; }
  void set_halfedge(Face f, Halfedge h) { fconn_[f.idx()].halfedge_ = h; }
  Halfedge halfedge(Edge e, unsigned i) const { return Halfedge((e.idx() << 1) + i); }
  Edge edge(Halfedge h) const { return Edge(static_cast<IndexType>(h.idx() >> 1)); }
  Vertex vertex(Edge e, unsigned i) const { return to_vertex(halfedge(e, i)); }
  Vertex to_vertex(Halfedge h) const { return hconn_[h.idx()].vertex_; }
  Vertex from_vertex(Halfedge h) const { return to_vertex(opposite_halfedge(h)); }
  void set_vertex(Halfedge h, Vertex v) { hconn_[h.idx()].vertex_ = v; }
  Face face(Halfedge h) const { return hconn_[h.idx()].face_; }
  Face face(Edge e, unsigned i) const { return face(halfedge(e, i)); }
  void set_face(Halfedge h, Face f) { hconn_[h.idx()].face_ = f; }
  Halfedge next_halfedge(Halfedge h) const { return hconn_[h.idx()].next_halfedge_; }
  Halfedge prev_halfedge(Halfedge h) const { return hconn_[h.idx()].prev_halfedge_; }
  void set_next_halfedge(Halfedge h, Halfedge nh) { hconn_[h.idx()].next_halfedge_ = nh; hconn_[nh.idx()].prev_halfedge_ = h; }
  Halfedge opposite_halfedge(Halfedge h) const { return Halfedge((h.idx() & 1) ? h.idx() - 1 : h.idx() + 1); }
  Halfedge ccw_rotated_halfedge(Halfedge h) const { return opposite_halfedge(prev_halfedge(h)); }
  Halfedge cw_rotated_halfedge(Halfedge h) const { return next_halfedge(opposite_halfedge(h)); }
  bool is_boundary(Halfedge h) const { return !face(h).is_valid(); }
  bool is_boundary(Vertex v) const { const Halfedge h = halfedge(v); return !(h.is_valid() && face(h).is_valid()); }
  bool is_isolated(Vertex v) const { return !halfedge(v).is_valid(); }

  Halfedge find_halfedge(Vertex start, Vertex end) const
  {
    Halfedge h = halfedge(start);
    const Halfedge hh = h;
    if (h.is_valid()) {
      do {
        if (to_vertex(h) == end) return h;
        h = cw_rotated_halfedge(h);
      } while (h != hh);
    }
    return Halfedge();
  }
  Edge find_edge(Vertex a, Vertex b) const { const Halfedge h = find_halfedge(a, b); return h.is_valid() ? edge(h) : Edge(); }

  Halfedge new_edge(Vertex start, Vertex end)
  {
    hconn_.emplace_back(); hconn_.emplace_back();
    const Halfedge h0(static_cast<IndexType>(hconn_.size() - 2));
    const Halfedge h1(static_cast<IndexType>(hconn_.size() - 1));
    set_vertex(h0, end);
    set_vertex(h1, start);
    return h0;
  }

  void adjust_outgoing_halfedge(Vertex v)
  {
    Halfedge h = halfedge(v);
    const Halfedge hh = h;
    if (h.is_valid()) {
      do {
        if (is_boundary(h)) { set_halfedge(v, h); return; }
        h = cw_rotated_halfedge(h);
      } while (h != hh);
    }
  }

  // pmp::SurfaceMesh::add_face (OpenMesh algorithm), triangles or general polygons
  Face add_face(const std::vector<Vertex>& vertices)
  {
    const std::size_t n = vertices.size();
    std::vector<Halfedge> halfedges(n);
    std::vector<bool> is_new(n), needs_adjust(n, false);
    std::vector<std::pair<Halfedge, Halfedge>> next_cache;
    next_cache.reserve(3 * n);
    Halfedge inner_next, inner_prev, outer_next, outer_prev, boundary_next, boundary_prev, patch_start, patch_end;
    std::size_t i, ii;

    for (i = 0, ii = 1; i < n; ++i, ++ii, ii %= n) {
      if (!is_boundary(vertices[i])) throw TopologyException("SurfaceMesh::add_face: Complex vertex.");
      halfedges[i] = find_halfedge(vertices[i], vertices[ii]);
      is_new[i] = !halfedges[i].is_valid();
      if (!is_new[i] && !is_boundary(halfedges[i])) throw TopologyException("SurfaceMesh::add_face: Complex edge.");
    }

    for (i = 0, ii = 1; i < n; ++i, ++ii, ii %= n) {
      if (!is_new[i] && !is_new[ii]) {
        inner_prev = halfedges[i];
        inner_next = halfedges[ii];
        if (next_halfedge(inner_prev) != inner_next) {
          outer_prev = opposite_halfedge(inner_next);
          outer_next = opposite_halfedge(inner_prev);
          boundary_prev = outer_prev;
          do {
            boundary_prev = opposite_halfedge(next_halfedge(boundary_prev));
          } while (!is_boundary(boundary_prev) || boundary_prev == inner_prev);
          boundary_next = next_halfedge(boundary_prev);
          if (boundary_next == inner_next) throw TopologyException("SurfaceMesh::add_face: Patch re-linking failed.");
          patch_start = next_halfedge(inner_prev);
          patch_end = prev_halfedge(inner_next);
          next_cache.emplace_back(boundary_prev, patch_start);
          next_cache.emplace_back(patch_end, boundary_next);
          next_cache.emplace_back(inner_prev, inner_next);
        }
      }
    }

    for (i = 0, ii = 1; i < n; ++i, ++ii, ii %= n)
      if (is_new[i]) halfedges[i] = new_edge(vertices[i], vertices[ii]);

    fconn_.emplace_back();
    const Face f(static_cast<IndexType>(fconn_.size() - 1));
    set_halfedge(f, halfedges[n - 1]);

    for (i = 0, ii = 1; i < n; ++i, ++ii, ii %= n) {
      const Vertex v = vertices[ii];
      inner_prev = halfedges[i];
      inner_next = halfedges[ii];
      unsigned id = 0;
      if (is_new[i]) id |= 1;
      if (is_new[ii]) id |= 2;
      if (id) {
        outer_prev = opposite_halfedge(inner_next);
        outer_next = opposite_halfedge(inner_prev);
        switch (id) {
          case 1:  // prev is new, next is old
            boundary_prev = prev_halfedge(inner_next);
            next_cache.emplace_back(boundary_prev, outer_next);
            set_halfedge(v, outer_next);
            break;
          case 2:  // next is new, prev is old
            boundary_next = next_halfedge(inner_prev);
            next_cache.emplace_back(outer_prev, boundary_next);
            set_halfedge(v, boundary_next);
            break;
          case 3:  // both are new
            if (!halfedge(v).is_valid()) {
              set_halfedge(v, outer_next);
              next_cache.emplace_back(outer_prev, outer_next);
            } else {
              boundary_next = halfedge(v);
              boundary_prev = prev_halfedge(boundary_next);
              next_cache.emplace_back(boundary_prev, outer_next);
              next_cache.emplace_back(outer_prev, boundary_next);
            }
            break;
        }
        next_cache.emplace_back(inner_prev, inner_next);
      } else {
        needs_adjust[ii] = (halfedge(v) == inner_next);
      }
      set_face(halfedges[i], f);
    }

    for (const auto& nc : next_cache) set_next_halfedge(nc.first, nc.second);
    for (i = 0; i < n; ++i)
      if (needs_adjust[i]) adjust_outgoing_halfedge(vertices[i]);
    return f;
  }

  // ---- circulators (CCW rotation, as pmp's *AroundVertexCirculator) ----
  class VertexAroundVertexRange
  {
  public:
    struct iterator
    {
      const SurfaceMesh* m; Halfedge h; bool active;
      Vertex operator*() const { return m->to_vertex(h); }
      Halfedge halfedge() const { return h; }
      iterator& operator++() { h = m->ccw_rotated_halfedge(h); active = true; return *this; }
      bool operator==(const iterator& o) const { return active && h == o.h; }
      bool operator!=(const iterator& o) const { return !(*this == o); }
    };
    VertexAroundVertexRange(const SurfaceMesh* m, Vertex v) : m_(m), h_(m->halfedge(v)) { }
    // pmp circulators are "loop once" iterators: begin()==end() compare unequal until the first increment
    iterator begin() const { return iterator{ m_, h_, !h_.is_valid() }; }
    iterator end() const { return iterator{ m_, h_, true }; }
  private:
    const SurfaceMesh* m_; Halfedge h_;
  };
  VertexAroundVertexRange vertices(Vertex v) const { return VertexAroundVertexRange(this, v); }

  // outgoing halfedges around v in circulator order (helper for the lvr2 wrapper)
  void halfedges_around(Vertex v, std::vector<Halfedge>& out) const
  {
    Halfedge h = halfedge(v);
    const Halfedge hh = h;
    if (!h.is_valid()) return;
    do { out.push_back(h); h = ccw_rotated_halfedge(h); } while (h != hh);
  }

private:
  struct VertexConnectivity { Halfedge halfedge_; };
  struct HalfedgeConnectivity { Face face_; Vertex vertex_; Halfedge next_halfedge_, prev_halfedge_; };
  struct FaceConnectivity { Halfedge halfedge_; };
  std::vector<VertexConnectivity> vconn_;
  std::vector<HalfedgeConnectivity> hconn_;
  std::vector<FaceConnectivity> fconn_;
};
}  // namespace pmp

// ======================================================================================================
// lvr2
// ======================================================================================================
namespace lvr2
{
using Index = std::uint32_t;
using VertexHandle = pmp::Vertex;
using EdgeHandle = pmp::Edge;
using FaceHandle = pmp::Face;
using HalfEdgeHandle = pmp::Halfedge;

class PanicException : public std::exception
{
public:
  PanicException(std::string msg = "") : msg_(std::move(msg)) { }
  const char* what() const noexcept override { return msg_.c_str(); }
private:
  std::string msg_;
};
class VertexLoopException : public std::exception
{
public:
  VertexLoopException(std::string msg = "") : msg_(std::move(msg)) { }
  const char* what() const noexcept override { return msg_.c_str(); }
private:
  std::string msg_;
};
[[noreturn]] inline void panic(const std::string& msg) { throw PanicException("Program panicked: " + msg); }

template <typename HandleT>
class OptionalHandle
{
public:
  OptionalHandle() : h_() { }
  OptionalHandle(HandleT h) : h_(h) { }
  OptionalHandle(boost::none_t) : h_() { }
  explicit operator bool() const { return h_.is_valid(); }
  bool operator!() const { return !h_.is_valid(); }
  HandleT unwrap() const { if (!h_.is_valid()) panic("unwrap on none optional handle"); return h_; }
  bool operator==(const OptionalHandle& o) const { return h_ == o.h_; }
  bool operator!=(const OptionalHandle& o) const { return h_ != o.h_; }
private:
  HandleT h_;
};
using OptionalVertexHandle = OptionalHandle<VertexHandle>;
using OptionalEdgeHandle = OptionalHandle<EdgeHandle>;
using OptionalFaceHandle = OptionalHandle<FaceHandle>;

// ------------------------------------------------------------------------------------------------------
// BaseVector / Normal
// ------------------------------------------------------------------------------------------------------
template <typename CoordT> struct Normal;

template <typename CoordT>
struct BaseVector
{
  using CoordType = CoordT;
  CoordT x, y, z;
  BaseVector() : x(0), y(0), z(0) { }
  BaseVector(const CoordT& x_, const CoordT& y_, const CoordT& z_) : x(x_), y(y_), z(z_) { }

  CoordT length2() const { return x * x + y * y + z * z; }
  CoordT length() const { return std::sqrt(length2()); }
  CoordT distance2(const BaseVector& o) const { return (*this - o).length2(); }
  CoordT distance(const BaseVector& o) const { return (*this - o).length(); }
  CoordT distanceFrom(const BaseVector& o) const { return distance(o); }
  CoordT dot(const BaseVector& o) const { return x * o.x + y * o.y + z * o.z; }
  BaseVector cross(const BaseVector& o) const { return BaseVector(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
  void normalize() { const CoordT l = length(); x /= l; y /= l; z /= l; }
  BaseVector normalized() const { BaseVector r(*this); r.normalize(); return r; }
  // CONVENTION (lvr2 source unavailable): Rodrigues rotation about the unit axis n by `angle` radians in
  // float: v' = v cos + (n x v) sin + n (n.v)(1 - cos)
  BaseVector rotated(const BaseVector& n, const double& angle) const
  {
    const CoordT a = static_cast<CoordT>(angle);
    const CoordT c = std::cos(a), s = std::sin(a);
    const BaseVector nxv = n.cross(*this);
    const CoordT k = n.dot(*this) * (CoordT(1) - c);
    return BaseVector(x * c + nxv.x * s + n.x * k, y * c + nxv.y * s + n.y * k, z * c + nxv.z * s + n.z * k);
  }

  BaseVector operator+(const BaseVector& o) const { return BaseVector(x + o.x, y + o.y, z + o.z); }
  BaseVector operator-(const BaseVector& o) const { return BaseVector(x - o.x, y - o.y, z - o.z); }
  BaseVector operator-() const { return BaseVector(-x, -y, -z); }
  BaseVector operator*(const CoordT& s) const { return BaseVector(x * s, y * s, z * s); }
  BaseVector operator/(const CoordT& s) const { return BaseVector(x / s, y / s, z / s); }
  BaseVector& operator+=(const BaseVector& o) { x += o.x; y += o.y; z += o.z; return *this; }
  BaseVector& operator-=(const BaseVector& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
  BaseVector& operator*=(const CoordT& s) { x *= s; y *= s; z *= s; return *this; }
  BaseVector& operator/=(const CoordT& s) { x /= s; y /= s; z /= s; return *this; }
  bool operator==(const BaseVector& o) const { return x == o.x && y == o.y && z == o.z; }
  bool operator!=(const BaseVector& o) const { return !(*this == o); }
  CoordT operator[](unsigned i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <typename CoordT> std::ostream& operator<<(std::ostream& os, const BaseVector<CoordT>& v)
{
  return os << '[' << v.x << ", " << v.y << ", " << v.z << ']';
}

// lvr2::Normal: a BaseVector that is normalised on construction
template <typename CoordT>
struct Normal : public BaseVector<CoordT>
{
  Normal() : BaseVector<CoordT>(0, 0, 1) { }
  Normal(const CoordT& x_, const CoordT& y_, const CoordT& z_) : BaseVector<CoordT>(x_, y_, z_) { this->normalize(); }
  Normal(const BaseVector<CoordT>& b) : BaseVector<CoordT>(b) { this->normalize(); }
  // bit-preserving construction for values that come out of a map file (no re-normalisation)
  static Normal raw(const CoordT& x_, const CoordT& y_, const CoordT& z_) { Normal n; n.x = x_; n.y = y_; n.z = z_; return n; }
};

using RGB8Color = std::array<std::uint8_t, 3>;

// ------------------------------------------------------------------------------------------------------
// attribute maps
// ------------------------------------------------------------------------------------------------------
template <typename HandleT, typename ValueT>
class AttributeMap
{
public:
  virtual ~AttributeMap() = default;
  virtual bool containsKey(HandleT key) const = 0;
  virtual boost::optional<ValueT> insert(HandleT key, const ValueT& value) = 0;
  virtual boost::optional<ValueT> erase(HandleT key) = 0;
  virtual void clear() = 0;
  virtual boost::optional<ValueT&> get(HandleT key) = 0;
  virtual boost::optional<const ValueT&> get(HandleT key) const = 0;
  virtual std::size_t numValues() const = 0;
  // next stored key with index >= pos, or SIZE_MAX
  virtual std::size_t nextKeyFrom(std::size_t pos) const = 0;

  ValueT& operator[](HandleT key)
  {
    auto m = get(key);
    if (!m) panic("attempt to access a non-existing value in an attribute map");
    return *m;
  }
  const ValueT& operator[](HandleT key) const
  {
    auto m = get(key);
    if (!m) panic("attempt to access a non-existing value in an attribute map");
    return *m;
  }

  struct iterator
  {
    const AttributeMap* m; std::size_t pos;
    HandleT operator*() const { return HandleT(static_cast<pmp::IndexType>(pos)); }
    iterator& operator++() { pos = m->nextKeyFrom(pos + 1); return *this; }
    bool operator!=(const iterator& o) const { return pos != o.pos; }
    bool operator==(const iterator& o) const { return pos == o.pos; }
  };
  iterator begin() const { return iterator{ this, nextKeyFrom(0) }; }
  iterator end() const { return iterator{ this, static_cast<std::size_t>(-1) }; }
};

// VectorMap on a StableVector: dense slots with a "used" flag, optional default value
template <typename HandleT, typename ValueT>
class VectorMap : public AttributeMap<HandleT, ValueT>
{
public:
  VectorMap() { }
  explicit VectorMap(const ValueT& defaultValue) : default_(defaultValue) { }
  VectorMap(std::size_t countElements, const ValueT& defaultValue)
    : slots_(countElements, Slot{ defaultValue, true }), used_(countElements), default_(defaultValue) { }

  bool containsKey(HandleT key) const override { return key.idx() < slots_.size() && slots_[key.idx()].used; }
  boost::optional<ValueT> insert(HandleT key, const ValueT& value) override
  {
    const std::size_t i = key.idx();
    if (i >= slots_.size()) { slots_.resize(i + 1); }
    Slot& s = slots_[i];
    boost::optional<ValueT> old;
    if (s.used) old = s.v; else { s.used = true; ++used_; }
    s.v = value;
    return old;
  }
  boost::optional<ValueT> erase(HandleT key) override
  {
    if (!containsKey(key)) return boost::none;
    Slot& s = slots_[key.idx()];
    boost::optional<ValueT> old(s.v);
    s.used = false; --used_;
    return old;
  }
  void clear() override { slots_.clear(); used_ = 0; }
  boost::optional<ValueT&> get(HandleT key) override
  {
    if (containsKey(key)) return slots_[key.idx()].v;
    if (default_) { insert(key, *default_); return slots_[key.idx()].v; }
    return boost::none;
  }
  boost::optional<const ValueT&> get(HandleT key) const override
  {
    if (containsKey(key)) return boost::optional<const ValueT&>(slots_[key.idx()].v);
    if (default_) return boost::optional<const ValueT&>(*default_);
    return boost::none;
  }
  std::size_t numValues() const override { return used_; }
  std::size_t nextKeyFrom(std::size_t pos) const override
  {
    for (; pos < slots_.size(); ++pos) if (slots_[pos].used) return pos;
    return static_cast<std::size_t>(-1);
  }
  void reserve(std::size_t n) { slots_.reserve(n); }
  using AttributeMap<HandleT, ValueT>::operator[];
private:
  struct Slot { ValueT v{}; bool used = false; };
  std::vector<Slot> slots_;
  std::size_t used_ = 0;
  boost::optional<ValueT> default_;
};

template <typename HandleT, typename ValueT>
class HashMap : public AttributeMap<HandleT, ValueT>
{
public:
  HashMap() { }
  explicit HashMap(const ValueT& defaultValue) : default_(defaultValue) { }
  bool containsKey(HandleT key) const override { return map_.count(key.idx()) > 0; }
  boost::optional<ValueT> insert(HandleT key, const ValueT& value) override
  {
    auto it = map_.find(key.idx());
    if (it != map_.end()) { boost::optional<ValueT> old(it->second); it->second = value; return old; }
    map_.emplace(key.idx(), value);
    return boost::none;
  }
  boost::optional<ValueT> erase(HandleT key) override
  {
    auto it = map_.find(key.idx());
    if (it == map_.end()) return boost::none;
    boost::optional<ValueT> old(it->second); map_.erase(it); return old;
  }
  void clear() override { map_.clear(); }
  boost::optional<ValueT&> get(HandleT key) override
  {
    auto it = map_.find(key.idx());
    if (it != map_.end()) return it->second;
    if (default_) { auto r = map_.emplace(key.idx(), *default_); return r.first->second; }
    return boost::none;
  }
  boost::optional<const ValueT&> get(HandleT key) const override
  {
    auto it = map_.find(key.idx());
    if (it != map_.end()) return boost::optional<const ValueT&>(it->second);
    if (default_) return boost::optional<const ValueT&>(*default_);
    return boost::none;
  }
  std::size_t numValues() const override { return map_.size(); }
  std::size_t nextKeyFrom(std::size_t pos) const override
  {
    auto it = map_.lower_bound(static_cast<pmp::IndexType>(pos));
    return it == map_.end() ? static_cast<std::size_t>(-1) : it->first;
  }
  void reserve(std::size_t) { }
private:
  std::map<pmp::IndexType, ValueT> map_;   // ordered: deterministic iteration
  boost::optional<ValueT> default_;
};

template <typename ValueT> using VertexMap = AttributeMap<VertexHandle, ValueT>;
template <typename ValueT> using EdgeMap = AttributeMap<EdgeHandle, ValueT>;
template <typename ValueT> using FaceMap = AttributeMap<FaceHandle, ValueT>;
template <typename ValueT> using DenseVertexMap = VectorMap<VertexHandle, ValueT>;
template <typename ValueT> using DenseEdgeMap = VectorMap<EdgeHandle, ValueT>;
template <typename ValueT> using DenseFaceMap = VectorMap<FaceHandle, ValueT>;
template <typename ValueT> using SparseVertexMap = HashMap<VertexHandle, ValueT>;
template <typename ValueT> using SparseEdgeMap = HashMap<EdgeHandle, ValueT>;
template <typename ValueT> using SparseFaceMap = HashMap<FaceHandle, ValueT>;
template <typename ValueT> using DenseVertexMapOptional = boost::optional<DenseVertexMap<ValueT>>;
template <typename ValueT> using DenseEdgeMapOptional = boost::optional<DenseEdgeMap<ValueT>>;
template <typename ValueT> using DenseFaceMapOptional = boost::optional<DenseFaceMap<ValueT>>;

// ------------------------------------------------------------------------------------------------------
// Meap: binary min-heap with key -> heap-index map (lvr2/util/Meap.{hpp,tcc})
// ------------------------------------------------------------------------------------------------------
template <typename KeyT, typename ValueT>
struct MeapPair
{
  KeyT key_; ValueT value_;
  const KeyT& key() const { return key_; }
  const ValueT& value() const { return value_; }
};

template <typename KeyT, typename ValueT>
class Meap
{
public:
  Meap() { }
  explicit Meap(std::size_t capacity) { heap_.reserve(capacity); }
  bool isEmpty() const { return heap_.empty(); }
  std::size_t numValues() const { return heap_.size(); }
  bool containsKey(KeyT key) const { return index_.count(key.idx()) > 0; }
  void clear() { heap_.clear(); index_.clear(); }

  boost::optional<ValueT> insert(const KeyT& key, const ValueT& value)
  {
    auto it = index_.find(key.idx());
    if (it != index_.end()) {
      const ValueT old = heap_[it->second].value_;
      updateValue(key, value);
      return old;
    }
    const std::size_t idx = heap_.size();
    heap_.push_back(MeapPair<KeyT, ValueT>{ key, value });
    index_[key.idx()] = idx;
    bubbleUp(idx);
    return boost::none;
  }
  void updateValue(const KeyT& key, const ValueT& newValue)
  {
    const std::size_t idx = index_.at(key.idx());
    if (newValue > heap_[idx].value_) { heap_[idx].value_ = newValue; bubbleDown(idx); }
    else if (newValue < heap_[idx].value_) { heap_[idx].value_ = newValue; bubbleUp(idx); }
  }
  const MeapPair<KeyT, ValueT>& peekMin() const { return heap_.front(); }
  MeapPair<KeyT, ValueT> popMin()
  {
    if (heap_.empty()) panic("attempt to peek at min in an empty heap");
    swapElems(0, heap_.size() - 1);
    const MeapPair<KeyT, ValueT> out = heap_.back();
    heap_.pop_back();
    index_.erase(out.key_.idx());
    if (!heap_.empty()) bubbleDown(0);
    return out;
  }
private:
  static std::size_t father(std::size_t c) { return (c - 1) / 2; }
  static std::size_t leftChild(std::size_t f) { return 2 * f + 1; }
  static std::size_t rightChild(std::size_t f) { return 2 * f + 2; }
  void swapElems(std::size_t a, std::size_t b)
  {
    if (a == b) return;
    std::swap(heap_[a], heap_[b]);
    index_[heap_[a].key_.idx()] = a;
    index_[heap_[b].key_.idx()] = b;
  }
  void bubbleUp(std::size_t idx)
  {
    while (idx != 0 && heap_[idx].value_ < heap_[father(idx)].value_) { swapElems(idx, father(idx)); idx = father(idx); }
  }
  void bubbleDown(std::size_t idx)
  {
    for (;;) {
      const bool hasLeft = leftChild(idx) < heap_.size();
      const bool hasRight = rightChild(idx) < heap_.size();
      if ((hasLeft && heap_[idx].value_ > heap_[leftChild(idx)].value_) ||
          (hasRight && heap_[idx].value_ > heap_[rightChild(idx)].value_)) {
        std::size_t smaller;
        if (!hasRight || heap_[leftChild(idx)].value_ < heap_[rightChild(idx)].value_) smaller = leftChild(idx);
        else smaller = rightChild(idx);
        swapElems(idx, smaller);
        idx = smaller;
      } else break;
    }
  }
  std::vector<MeapPair<KeyT, ValueT>> heap_;
  std::unordered_map<pmp::IndexType, std::size_t> index_;
};

// ------------------------------------------------------------------------------------------------------
// MeshBuffer / Channel
// ------------------------------------------------------------------------------------------------------
template <typename T>
class Channel
{
public:
  Channel() : n_(0), w_(0) { }
  Channel(std::size_t n, std::size_t width) : n_(n), w_(width), data_(new T[n * width]()) { }
  std::size_t numElements() const { return n_; }
  std::size_t width() const { return w_; }
  T* operator[](std::size_t i) { return data_.get() + i * w_; }
  const T* operator[](std::size_t i) const { return data_.get() + i * w_; }
  std::shared_ptr<T[]> dataPtr() const { return data_; }
private:
  std::size_t n_, w_;
  std::shared_ptr<T[]> data_;
};

class MeshBuffer
{
public:
  struct ChannelSlot
  {
    std::any value;
    template <typename T> ChannelSlot& operator=(const Channel<T>& c) { value = c; return *this; }
  };
  ChannelSlot& operator[](const std::string& name) { return channels_[name]; }
  template <typename T> boost::optional<Channel<T>> getChannel(const std::string& name) const
  {
    auto it = channels_.find(name);
    if (it == channels_.end()) return boost::none;
    if (const auto* c = std::any_cast<Channel<T>>(&it->second.value)) return *c;
    return boost::none;
  }
  std::size_t numVertices() const { auto c = getChannel<float>("vertices"); return c ? c->numElements() : 0; }
  std::size_t numFaces() const { auto c = getChannel<unsigned int>("face_indices"); return c ? c->numElements() : 0; }
  bool hasVertexColors() const { return channels_.count("vertex_colors") > 0; }
  std::shared_ptr<unsigned char[]> getVertexColors(std::size_t& width)
  {
    auto c = getChannel<unsigned char>("vertex_colors");
    if (!c) { width = 0; return nullptr; }
    width = c->width(); return c->dataPtr();
  }
  void setVertices(const float* xyz, std::size_t n)
  {
    Channel<float> c(n, 3);
    std::copy(xyz, xyz + 3 * n, c[0]);
    (*this)["vertices"] = c;
  }
  void setFaceIndices(const unsigned int* f, std::size_t n)
  {
    Channel<unsigned int> c(n, 3);
    std::copy(f, f + 3 * n, c[0]);
    (*this)["face_indices"] = c;
  }
private:
  std::map<std::string, ChannelSlot> channels_;
};
using MeshBufferPtr = std::shared_ptr<MeshBuffer>;
inline std::ostream& operator<<(std::ostream& os, const MeshBuffer& b)
{
  return os << "MeshBuffer(" << b.numVertices() << " vertices, " << b.numFaces() << " faces)";
}

// ------------------------------------------------------------------------------------------------------
// PMPMesh<BaseVecT>: lvr2's BaseMesh interface on top of pmp::SurfaceMesh
// ------------------------------------------------------------------------------------------------------
template <typename HandleT>
class HandleRange
{
public:
  explicit HandleRange(std::size_t n) : n_(n) { }
  struct iterator
  {
    std::size_t i;
    HandleT operator*() const { return HandleT(static_cast<pmp::IndexType>(i)); }
    iterator& operator++() { ++i; return *this; }
    bool operator!=(const iterator& o) const { return i != o.i; }
    bool operator==(const iterator& o) const { return i == o.i; }
  };
  iterator begin() const { return iterator{ 0 }; }
  iterator end() const { return iterator{ n_ }; }
private:
  std::size_t n_;
};

template <typename BaseVecT>
class PMPMesh
{
public:
  PMPMesh() { }
  explicit PMPMesh(MeshBufferPtr buffer)
  {
    const auto verts = buffer->getChannel<float>("vertices");
    const auto faces = buffer->getChannel<unsigned int>("face_indices");
    if (verts) for (std::size_t i = 0; i < verts->numElements(); ++i) addVertex(BaseVecT((*verts)[i][0], (*verts)[i][1], (*verts)[i][2]));
    if (faces) {
      for (std::size_t i = 0; i < faces->numElements(); ++i) {
        try { addFace(VertexHandle((*faces)[i][0]), VertexHandle((*faces)[i][1]), VertexHandle((*faces)[i][2])); }
        catch (const pmp::TopologyException&) { /* non-manifold face skipped, like lvr2 */ }
      }
    }
  }

  VertexHandle addVertex(BaseVecT pos) { positions_.push_back(pos); return mesh_.add_vertex_slot(); }
  FaceHandle addFace(VertexHandle a, VertexHandle b, VertexHandle c) { return mesh_.add_face({ a, b, c }); }

  std::size_t numVertices() const { return mesh_.n_vertices(); }
  std::size_t numFaces() const { return mesh_.n_faces(); }
  std::size_t numEdges() const { return mesh_.n_edges(); }
  Index nextVertexIndex() const { return static_cast<Index>(mesh_.n_vertices()); }
  Index nextFaceIndex() const { return static_cast<Index>(mesh_.n_faces()); }
  Index nextEdgeIndex() const { return static_cast<Index>(mesh_.n_edges()); }
  bool containsVertex(VertexHandle v) const { return v.idx() < mesh_.n_vertices(); }
  bool containsFace(FaceHandle f) const { return f.idx() < mesh_.n_faces(); }
  bool containsEdge(EdgeHandle e) const { return e.idx() < mesh_.n_edges(); }

  HandleRange<VertexHandle> vertices() const { return HandleRange<VertexHandle>(mesh_.n_vertices()); }
  HandleRange<FaceHandle> faces() const { return HandleRange<FaceHandle>(mesh_.n_faces()); }
  HandleRange<EdgeHandle> edges() const { return HandleRange<EdgeHandle>(mesh_.n_edges()); }

  const BaseVecT& getVertexPosition(VertexHandle v) const { check(v); return positions_[v.idx()]; }
  BaseVecT& getVertexPosition(VertexHandle v) { check(v); return positions_[v.idx()]; }

  std::array<VertexHandle, 3> getVerticesOfFace(FaceHandle f) const
  {
    pmp::Halfedge h = mesh_.halfedge(f);
    const VertexHandle a = mesh_.to_vertex(h); h = mesh_.next_halfedge(h);
    const VertexHandle b = mesh_.to_vertex(h); h = mesh_.next_halfedge(h);
    const VertexHandle c = mesh_.to_vertex(h);
    return { a, b, c };
  }
  std::array<BaseVecT, 3> getVertexPositionsOfFace(FaceHandle f) const
  {
    const auto v = getVerticesOfFace(f);
    return { positions_[v[0].idx()], positions_[v[1].idx()], positions_[v[2].idx()] };
  }
  std::array<VertexHandle, 2> getVerticesOfEdge(EdgeHandle e) const { return { mesh_.vertex(e, 0), mesh_.vertex(e, 1) }; }
  std::array<OptionalFaceHandle, 2> getFacesOfEdge(EdgeHandle e) const
  {
    const FaceHandle f0 = mesh_.face(e, 0), f1 = mesh_.face(e, 1);
    return { f0.is_valid() ? OptionalFaceHandle(f0) : OptionalFaceHandle(), f1.is_valid() ? OptionalFaceHandle(f1) : OptionalFaceHandle() };
  }

  void getEdgesOfVertex(VertexHandle v, std::vector<EdgeHandle>& out) const
  {
    check(v);
    std::vector<pmp::Halfedge> hs; mesh_.halfedges_around(v, hs);
    for (auto h : hs) out.push_back(mesh_.edge(h));
  }
  std::vector<EdgeHandle> getEdgesOfVertex(VertexHandle v) const { std::vector<EdgeHandle> o; getEdgesOfVertex(v, o); return o; }
  void getFacesOfVertex(VertexHandle v, std::vector<FaceHandle>& out) const
  {
    check(v);
    std::vector<pmp::Halfedge> hs; mesh_.halfedges_around(v, hs);
    for (auto h : hs) { const FaceHandle f = mesh_.face(h); if (f.is_valid()) out.push_back(f); }
  }
  std::vector<FaceHandle> getFacesOfVertex(VertexHandle v) const { std::vector<FaceHandle> o; getFacesOfVertex(v, o); return o; }
  void getNeighboursOfVertex(VertexHandle v, std::vector<VertexHandle>& out) const
  {
    std::vector<pmp::Halfedge> hs; mesh_.halfedges_around(v, hs);
    for (auto h : hs) out.push_back(mesh_.to_vertex(h));
  }

  OptionalEdgeHandle getEdgeBetween(VertexHandle a, VertexHandle b) const
  {
    const pmp::Edge e = mesh_.find_edge(a, b);
    return e.is_valid() ? OptionalEdgeHandle(e) : OptionalEdgeHandle();
  }
  // lvr2 BaseMesh::getFaceBetween: one of the two faces of edge (a,b) that also contains c
  OptionalFaceHandle getFaceBetween(VertexHandle a, VertexHandle b, VertexHandle c) const
  {
    const auto e = getEdgeBetween(a, b);
    if (!e) return OptionalFaceHandle();
    for (const auto& of : getFacesOfEdge(e.unwrap())) {
      if (!of) continue;
      const auto vs = getVerticesOfFace(of.unwrap());
      if (vs[0] == c || vs[1] == c || vs[2] == c) return of;
    }
    return OptionalFaceHandle();
  }

  const pmp::SurfaceMesh& getSurfaceMesh() const { return mesh_; }
  pmp::SurfaceMesh& getSurfaceMesh() { return mesh_; }

private:
  void check(VertexHandle v) const { if (v.idx() >= positions_.size()) panic("access to a vertex that is not part of the mesh"); }
  pmp::SurfaceMesh mesh_;
  std::vector<BaseVecT> positions_;
};

// SimpleFinalizer: re-export a mesh into a buffer with contiguous indices
template <typename BaseVecT>
class SimpleFinalizer
{
public:
  void setColorData(const DenseVertexMap<RGB8Color>&) { }
  MeshBufferPtr apply(const PMPMesh<BaseVecT>& mesh)
  {
    auto buf = std::make_shared<MeshBuffer>();
    Channel<float> v(mesh.numVertices(), 3);
    for (std::size_t i = 0; i < mesh.numVertices(); ++i) { const auto& p = mesh.getVertexPosition(VertexHandle(i)); v[i][0] = p.x; v[i][1] = p.y; v[i][2] = p.z; }
    Channel<unsigned int> f(mesh.numFaces(), 3);
    for (std::size_t i = 0; i < mesh.numFaces(); ++i) { const auto vs = mesh.getVerticesOfFace(FaceHandle(i)); for (int k = 0; k < 3; ++k) f[i][k] = vs[k].idx(); }
    (*buf)["vertices"] = v; (*buf)["face_indices"] = f;
    return buf;
  }
};

// ------------------------------------------------------------------------------------------------------
// algorithms (lvr2/algorithm/NormalAlgorithms.tcc, GeometryAlgorithms.tcc)
// ------------------------------------------------------------------------------------------------------
template <typename BaseVecT>
boost::optional<Normal<typename BaseVecT::CoordType>> getFaceNormal(std::array<BaseVecT, 3> face)
{
  const auto v1 = face[0], v2 = face[1], v3 = face[2];
  const auto normal = (v1 - v2).cross(v1 - v3);
  if (normal.length2() == 0) return boost::none;
  return Normal<typename BaseVecT::CoordType>(normal);
}
template <typename BaseVecT>
DenseFaceMap<Normal<typename BaseVecT::CoordType>> calcFaceNormals(const PMPMesh<BaseVecT>& mesh)
{
  DenseFaceMap<Normal<typename BaseVecT::CoordType>> out;
  out.reserve(mesh.nextFaceIndex());
  for (auto fH : mesh.faces()) {
    auto maybe = getFaceNormal(mesh.getVertexPositionsOfFace(fH));
    out.insert(fH, maybe ? *maybe : Normal<typename BaseVecT::CoordType>(0, 0, 1));
  }
  return out;
}
template <typename BaseVecT>
DenseVertexMap<Normal<typename BaseVecT::CoordType>> calcVertexNormals(
    const PMPMesh<BaseVecT>& mesh, const FaceMap<Normal<typename BaseVecT::CoordType>>& normals)
{
  DenseVertexMap<Normal<typename BaseVecT::CoordType>> out;
  out.reserve(mesh.nextVertexIndex());
  for (auto vH : mesh.vertices()) {
    const auto faces = mesh.getFacesOfVertex(vH);
    if (faces.empty()) continue;
    BaseVecT v(0, 0, 0);
    for (auto f : faces) v += normals[f];
    out.insert(vH, Normal<typename BaseVecT::CoordType>(v));
  }
  return out;
}
template <typename BaseVecT>
DenseEdgeMap<float> calcVertexDistances(const PMPMesh<BaseVecT>& mesh)
{
  DenseEdgeMap<float> out;
  out.reserve(mesh.nextEdgeIndex());
  for (auto eH : mesh.edges()) {
    const auto vs = mesh.getVerticesOfEdge(eH);
    out.insert(eH, mesh.getVertexPosition(vs[0]).distanceFrom(mesh.getVertexPosition(vs[1])));
  }
  return out;
}

// ------------------------------------------------------------------------------------------------------
// map-file IO: an in-memory stand-in for the HDF5 attribute store.  A "file" is a named store the
// harness fills before MeshMap::readMap() runs (ref_store_open).
// ------------------------------------------------------------------------------------------------------
struct RefStore
{
  std::map<std::string, MeshBufferPtr> meshes;
  std::map<std::string, std::any> attributes;   // key: mesh name + "/" + attribute name
};
inline std::map<std::string, std::shared_ptr<RefStore>>& ref_store_registry()
{
  static std::map<std::string, std::shared_ptr<RefStore>> r;
  return r;
}
inline std::shared_ptr<RefStore> ref_store_open(const std::string& file)
{
  auto& r = ref_store_registry();
  auto it = r.find(file);
  if (it == r.end()) it = r.emplace(file, std::make_shared<RefStore>()).first;
  return it->second;
}

class AttributeMeshIOBase
{
public:
  virtual ~AttributeMeshIOBase() = default;
  template <typename MapT> boost::optional<MapT> getDenseAttributeMap(const std::string& name)
  {
    if (!store_) return boost::none;
    auto it = store_->attributes.find(mesh_name_ + "/" + name);
    if (it == store_->attributes.end()) return boost::none;
    if (const auto* m = std::any_cast<MapT>(&it->second)) return *m;
    return boost::none;
  }
  template <typename MapT> boost::optional<MapT> getAttributeMap(const std::string& name) { return getDenseAttributeMap<MapT>(name); }
  template <typename MapT> bool addDenseAttributeMap(const MapT& map, const std::string& name)
  {
    if (!store_) return false;
    store_->attributes[mesh_name_ + "/" + name] = map;
    return true;
  }
  template <typename MapT> bool addAttributeMap(const MapT& map, const std::string& name) { return addDenseAttributeMap(map, name); }
protected:
  std::shared_ptr<RefStore> store_;
  std::string mesh_name_;
};

namespace hdf5features
{
class MeshIO
{
public:
  MeshBufferPtr load(const std::string& name)
  {
    if (!io_store_) return nullptr;
    auto it = io_store_->meshes.find(name);
    return it == io_store_->meshes.end() ? nullptr : it->second;
  }
  void save(const std::string& name, const MeshBufferPtr& buffer) { if (io_store_) io_store_->meshes[name] = buffer; }
protected:
  std::shared_ptr<RefStore> io_store_;
};
}  // namespace hdf5features

template <typename Feature>
class Hdf5Build : public AttributeMeshIOBase, public Feature
{
public:
  void open(const std::string& file) { store_ = ref_store_open(file); this->io_store_ = store_; }
  void setMeshName(const std::string& name) { mesh_name_ = name; }
};

// ------------------------------------------------------------------------------------------------------
// ray casting (only constructed, never queried on this path)
// ------------------------------------------------------------------------------------------------------
namespace intelem { struct Distance { }; struct Face { }; struct Point { }; }
template <typename... Elems> struct Intersection { };
template <typename ResultT> class RaycasterBase { public: virtual ~RaycasterBase() = default; };
template <typename ResultT> class BVHRaycaster : public RaycasterBase<ResultT> { public: explicit BVHRaycaster(MeshBufferPtr) { } };
template <typename ResultT> class EmbreeRaycaster : public RaycasterBase<ResultT> { public: explicit EmbreeRaycaster(MeshBufferPtr) { } };
}  // namespace lvr2
