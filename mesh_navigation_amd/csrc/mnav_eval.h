// mnav_eval.h -- per-vertex evaluation rules and band controller of the wavefront engine.
//
// Shared between the HIP kernels (mnav_kernels.hip) and the CPU schedule model that the
// test-suite uses to check the *schedule* against the sequential oracle (oracle/
// schedule_model.cpp, test infrastructure).  No HIP intrinsics in here; atomics and list
// pushes are injected by the caller.
//
// Engine in one paragraph: both reference planners are ordered wavefronts
// driven by a priority queue (dijkstra_mesh_planner.cpp:287-348, cvp_mesh_planner.cpp:747-886).
// We replace the queue by distance *bands*: all vertices whose pop time falls in
// [thr_fixed, thr) are settled together by iterating a per-vertex GATHER rule to its fixed
// point, then the band advances.  The gather rule recomputes a vertex from scratch out of the
// state of its neighbours and reproduces what the sequential loop would have done to that
// vertex: for Dijkstra a min over expanded neighbours (order-free), for CVP a replay of the
// incident-face updates in the order in which their triggering vertices pop.
#pragma once
#include <math.h>
#include <stdint.h>
#ifdef MNAV_CHECK_TWO_PART
#include <stdio.h>
#include <stdlib.h>
#endif

#if defined(__HIPCC__)
#define MNAV_HD __host__ __device__ __forceinline__
#define MNAV_HD_COLD __host__ __device__ __attribute__((noinline))   // rare paths: kept out of the callers' register budget
#define MNAV_UNROLL _Pragma("unroll")
#else
#define MNAV_HD inline
#define MNAV_HD_COLD inline
#define MNAV_UNROLL
#endif

namespace mnav {

constexpr uint32_t kNone = 0xFFFFFFFFu;

// MBF GetPath result codes, dijkstra_mesh_planner.h:72-85
enum : uint32_t { kSuccess = 0, kCanceled = 51, kInvalidStart = 52, kInvalidGoal = 53, kNoPathFound = 54,
                  kInternalError = 60 };   // INVALID_PLUGIN is 59

enum : uint32_t { kPlannerDijkstra = 0, kPlannerCvp = 1 };

// One directed CSR entry of the SSSP gather graph: row v holds {u, w(u,v)} for every
// neighbour u.  w is +inf when u may never act as a source for v (v invalid, or
// vertex_costs[u] > cost_limit: dijkstra_mesh_planner.cpp:302,328).
struct Nbr { uint32_t u; float w; };

// One incident-face corner of vertex v3 (CVP).  (v1,v2,v3) is the cyclic rotation of the
// face with v3 last (cvp_mesh_planner.cpp:811,834,857); a=w(v2,v3), b=w(v1,v3), c=w(v1,v2)
// (cvp_mesh_planner.cpp:380-390).  v1 == kNone marks a face that must be skipped because one
// of its vertices is invalid (cvp_mesh_planner.cpp:785).
// `face` carries the face id in its low 28 bits and order flags: the reference applies the faces of a
// popped vertex t in the order of t's half-edge circulator (getFacesOfVertex, cvp :775-778), and on
// cost-inflated triangles the update is not a pure minimum, so when both faces of edge (t, v3) fire on the
// same pop their order matters.  kCornerFirst1 / kCornerFirst2: this face comes first of the two in the
// circulator of v1 / v2.
struct Corner { uint32_t v1, v2; float a, b, c; uint32_t face; };
constexpr uint32_t kCornerFaceMask = 0x0FFFFFFFu, kCornerFirst1 = 0x40000000u, kCornerFirst2 = 0x80000000u;
// The inflation wave visits the faces of a popped vertex in another order than getFacesOfVertex: for every
// neighbour nh of the vertex circulator, the face left of cur->nh and then the face left of nh->cur
// (inflation_layer.cpp:423-427), so the LAST face of the circulator is met right after (interior vertex) or before
// (boundary vertex) the first one.  Its update is a minimum, but which update RE-QUEUES the vertex (:311) depends on
// the order.  The topology carries these flags in bits 28/29; the inflation corner table moves them to 30/31.
constexpr uint32_t kCornerInfl1 = 0x10000000u, kCornerInfl2 = 0x20000000u;
MNAV_HD uint32_t corner_face(const Corner& k) { return k.face & kCornerFaceMask; }
MNAV_HD bool corner_first_for(const Corner& k, uint32_t trig) { return (k.face & (trig == k.v1 ? kCornerFirst1 : kCornerFirst2)) != 0u; }
MNAV_HD uint32_t corner_face_for_inflation(uint32_t face) { return (face & kCornerFaceMask) | ((face & (kCornerInfl1 | kCornerInfl2)) << 2); }

MNAV_HD float u2f(uint32_t u) { union { uint32_t u; float f; } x; x.u = u; return x.f; }
MNAV_HD uint32_t f2u(float f) { union { uint32_t u; float f; } x; x.f = f; return x.u; }

// acosf as the reference's host computes it (SteepnessLayer, steepness_layer.cpp:165: `acos(normal.z)` on a float).  The
// device's own acosf differs from glibc's in the last bit for some arguments (seen on the 1M terrain), so the float
// algorithm glibc 2.35 uses (sysdeps/ieee754/flt-32/e_acosf.c, the fdlibm routine) is restated here operation by
// operation: plain IEEE float arithmetic with correctly rounded division and square root, no contraction.  Checked
// against the host's libm on all 2 130 706 434 floats of [-1, 1]: identical bits (tests/test_oracle_kat.py samples it).
MNAV_HD float acosf_ref(float x)
{
  const float one = 1.0000000000e+00f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
              pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
              pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
              qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  const uint32_t hx = f2u(x), ix = hx & 0x7fffffffu;
  if (ix == 0x3f800000u) return (hx >> 31) ? pi + 2.0f * pio2_lo : 0.0f;        // |x| == 1
  if (ix > 0x3f800000u) return (x - x) / (x - x);                                // |x| > 1 or NaN: NaN
  if (ix < 0x3f000000u) {                                                        // |x| < 0.5
    if (ix <= 0x23000000u) return pio2_hi + pio2_lo;
    const float z = x * x;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (hx >> 31) {                                                                // x < -0.5
    const float z = (one + x) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float s = sqrtf(z);
    const float r = p / q;
    const float w = r * s - pio2_lo;
    return pi - 2.0f * (s + w);
  }
  const float z = (one - x) * 0.5f;                                              // x > 0.5
  const float s = sqrtf(z);
  const float df = u2f(f2u(s) & 0xfffff000u);
  const float c = (z - df * df) / (s + df);
  const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const float r = p / q;
  const float w = r * s + c;
  return 2.0f * (df + w);
}
// cosf as the reference's host computes it (InflationLayer::vectorAt, inflation_layer.cpp:509: `cos(alpha)` on a float,
// read by MeshMap::meshAhead while the CVP planner walks its path).  glibc 2.35's routine (sysdeps/ieee754/flt-32/s_cosf.c
// with s_sincosf.h: quadrant reduction and two degree-8 polynomials in double, the x86-64 build with fused multiply-adds)
// restated operation by operation.  Checked against this host's libm on every float of [0, 120): identical bits; the
// variant without fused operations (a host without FMA) differs on 11 arguments, all beyond 17 rad -- outside what
// vectorAt can produce (alpha <= (sqrt(inflation_radius) - inscribed_radius) / (inflation_radius - inscribed_radius) * pi).
// |x| >= 120, infinities and NaNs are not restated (the caller never produces them): they return NaN.
namespace sincosf_ref_detail {
// s_sincosf.h: sinf_poly (n even: sine of x, odd: cosine; `flip` = the second table entry, whose cosine coefficients are negated)
MNAV_HD float poly(double x, double x2, int n, double flip)
{
  const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    const double x3 = x * x2, t1 = fma(x2, s3, s2), x7 = x3 * x2, s = fma(x3, s1, x);
    return (float)fma(x7, t1, s);
  }
  const double x4 = x2 * x2, q2 = fma(x2, flip * c4, flip * c3), q1 = fma(x2, flip * c1, flip * c0), x6 = x4 * x2, c = fma(x4, flip * c2, q1);
  return (float)fma(x6, q2, c);
}
// reduce_fast + the sign / table selection of s_sinf.c / s_cosf.c; `odd` = 1 for the cosine (n ^ 1)
MNAV_HD float reduced(double x, int odd)
{
  const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
  const double r = x * hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;                      // quadrant, rounded to nearest
  x = fma(-(double)n, hpi, x);
  const double sgn = (((n & 3) == 1) || ((n & 3) == 2)) ? -1.0 : 1.0;   // sign table {1, -1, -1, 1}
  return poly(x * sgn, x * x, n ^ odd, (n & 2) ? -1.0 : 1.0);
}
}  // namespace sincosf_ref_detail

MNAV_HD float cosf_ref(float y)
{
  const uint32_t top = (f2u(y) >> 20) & 0x7ffu;
  const double x = (double)y;
  if (top < ((0x3f490fdbu >> 20) & 0x7ffu)) {                       // |y| below pi/4 (compared on the top 12 bits, as s_cosf.c does)
    if (top < ((0x39800000u >> 20) & 0x7ffu)) return 1.0f;          // |y| < 2^-12
    return sincosf_ref_detail::poly(x, x * x, 1, 1.0);
  }
  if (!(top < ((0x42f00000u >> 20) & 0x7ffu))) return u2f(0x7fc00000u);
  return sincosf_ref_detail::reduced(x, 1);
}
// sinf likewise (s_sinf.c; CVPMeshPlanner's vector map rotates by the cut-face angle, cvp_mesh_planner.cpp:234): identical
// bits to this host's libm on every float of (-120, 120), checked the same way.
MNAV_HD float sinf_ref(float y)
{
  const uint32_t top = (f2u(y) >> 20) & 0x7ffu;
  const double x = (double)y;
  if (top < ((0x3f490fdbu >> 20) & 0x7ffu)) {
    if (top < ((0x39800000u >> 20) & 0x7ffu)) return y;
    return sincosf_ref_detail::poly(x, x * x, 0, 1.0);
  }
  if (!(top < ((0x42f00000u >> 20) & 0x7ffu))) return u2f(0x7fc00000u);
  return sincosf_ref_detail::reduced(x, 0);
}
MNAV_HD float inf_f() { return u2f(0x7f800000u); }
MNAV_HD float next_up(float x) { return (x >= 0.0f) ? u2f(f2u(x) + 1u) : u2f(f2u(x) - 1u); }  // finite x

// Total order of pops.  The reference pops the heap minimum, ties to the smaller vertex id (the tie rule
// fixed for the un-vendored lvr2::Meap, see DESIGN.md "tie rule").  A face update may set a vertex
// BELOW the value that is popping (non-causal update: obtuse / cost-inflated triangles, and above all
// waves that wrap around holes and obstacles and fill the shadow behind them backwards); that vertex
// is then the heap minimum and pops next, and it can start a whole CASCADE of pops below the main
// front -- which among themselves pop in (value, id) order again, and nest: a vertex set below the
// value popping INSIDE a cascade opens a sub-cascade, and so on, as deep as the backward wave runs.
// The pop order is therefore the PREORDER of a forest: roots = the pops of the main front in
// (value, id) order; the children of a node = the vertices that pop inside its (sub-)cascade, i.e.
// below its value but above that of any deeper node they could belong to, in (value, id) order.
// A pop key stores a node's place in that forest without copying the path:
//     hi  = [ float bits of the root's pop value : 32 | root id : 26 | 0 : 6 ]   (the main-front pop)
//     up  = parent node (vertex id), kNone for a root;   lvl = depth in the tree (0 for a root)
// the node's own (value, id) pair is (dist[v], v).  Keys of different roots compare by `hi` alone (one
// integer compare, the common case); inside one cascade key_less() walks the `up` links to the
// siblings that decide.  A dependent always sorts after the pop that set it, which keeps the gather
// iteration well-founded.  Limits: V <= 2^26 for CVP.
struct PopKey { unsigned long long hi; uint32_t up; uint32_t lvl; };
MNAV_HD bool operator==(const PopKey& a, const PopKey& b) { return a.hi == b.hi && a.up == b.up && a.lvl == b.lvl; }
MNAV_HD bool operator!=(const PopKey& a, const PopKey& b) { return !(a == b); }
constexpr uint32_t kKeyIdBits = 26, kKeyPadBits = 6;
constexpr int kKeyWalkMax = 4096;        // default bound on the `up` walks (a half-converged tree may be inconsistent); Plan.walk_max
MNAV_HD unsigned long long key_pair(float t, uint32_t id)
{
  return ((unsigned long long)f2u(t) << 32) | ((unsigned long long)(id & ((1u << kKeyIdBits) - 1u)) << kKeyPadBits);
}
MNAV_HD uint32_t pair_id(unsigned long long pr) { return (uint32_t)(pr >> kKeyPadBits) & ((1u << kKeyIdBits) - 1u); }
// key of an ordinary (main front) pop of vertex id at value t
MNAV_HD PopKey make_key(float t, uint32_t id)
{
  PopKey k; k.hi = key_pair(t, id); k.up = kNone; k.lvl = 0u;
  return k;
}
MNAV_HD float key_time(const PopKey& k) { return u2f((uint32_t)(k.hi >> 32)); }   // pop time on the main front (bands)
MNAV_HD PopKey key_inf() { return make_key(inf_f(), 0); }
// A key together with the node's own (value, id) pair -- what comparisons work on.
struct KeyRef { PopKey k; unsigned long long own; };
MNAV_HD KeyRef key_ref_of(const PopKey& k, float d, uint32_t v) { KeyRef r; r.k = k; r.own = key_pair(d, v); return r; }


// ---------------------------------------------------------------------------------------
// CVP triangle update, cvp_mesh_planner.cpp:369-556, on plain numbers.  float64 arithmetic,
// float32 result, no FMA contraction (build with -ffp-contract=off).
// ---------------------------------------------------------------------------------------
struct CvpUpd { float u3; float dir; int sel; bool ok; };  // sel: 1 -> pred=v1, 2 -> pred=v2

// The update splits into a part that does not depend on the current value u3 of the free vertex
// (candidate) and two strict '<' gates against u3 (apply).  kind: 1 = planar solution u3tmp is
// accepted as is (:493-517), 2 = edge fall-back u1+b / u2+a (:418-454, :518-553).
struct CvpCand { double u3tmp; double cand; float dir; int sel; int kind; };

MNAV_HD CvpCand cvp_candidate(float u1f, float u2f_, float af, float bf, float cf)
{
  CvpCand r;
  const double u1 = u1f, u2 = u2f_;                         // :376-377
  const double c = cf, c_sq = c * c;                        // :381-382
  const double b = bf, b_sq = b * b;                        // :385-386
  const double a = af, a_sq = a * a;                        // :389-390
  const double u1_sq = u1 * u1, u2_sq = u2 * u2;            // :392-393
  const double sx = (c_sq + u1_sq - u2_sq) / (2 * c);       // :395
  const double sy = -sqrt(fmax(u1_sq - sx * sx, 0.0));      // :396
  const double p = (b_sq + c_sq - a_sq) / (2 * c);          // :398
  const double hc = sqrt(fmax(b_sq - p * p, 0.0));          // :399
  const double dy = hc - sy, dx = p - sx;                   // :401-402
  const double u3tmp_sq = dx * dx + dy * dy;                // :404
  const double u3tmp = sqrt(u3tmp_sq);                      // :405
  r.u3tmp = u3tmp; r.cand = u3tmp; r.dir = 0.0f; r.sel = 0; r.kind = 1;
  const double t0a = (a_sq + b_sq - c_sq) / (2 * a * b);            // :413
  const double t1a = (u3tmp_sq + b_sq - u1_sq) / (2 * u3tmp * b);   // :414
  const double t2a = (a_sq + u3tmp_sq - u2_sq) / (2 * a * u3tmp);   // :415
  int fallback = 0;                                         // 1 -> u1+b, 2 -> u2+a
  if (fabs(t1a) > 1) fallback = 1;                          // :418
  else if (fabs(t2a) > 1) fallback = 2;                     // :437
  else {
    // The reference compares theta_k = acos(t_k) (:456-458, :493, :498, :518).  acos is strictly decreasing with
    // |slope| >= 1, so cosines that differ by more than kAcosWindow give angles that differ by more than that -- far
    // beyond the rounding of any acos -- and the comparison is decided on the cosines; only inside the window are the
    // angles computed and compared as the reference does.  Saves 2-3 of the 3 float64 acos per face update.
    constexpr double kAcosWindow = 1e-14;
    const double d10 = t1a - t0a, d20 = t2a - t0a, d12 = t1a - t2a;
    const int c10 = d10 > kAcosWindow ? 1 : (d10 < -kAcosWindow ? 0 : -1);   // theta1 < theta0 ?
    const int c20 = d20 > kAcosWindow ? 1 : (d20 < -kAcosWindow ? 0 : -1);   // theta2 < theta0 ?
    const int c12 = d12 > kAcosWindow ? 1 : (d12 < -kAcosWindow ? 0 : -1);   // theta1 < theta2 ?
    if (fabs(t0a) <= 1 && c10 >= 0 && c20 >= 0 && c12 >= 0) {         // (t0a is not range-checked by the reference: acos -> NaN)
      if (c10 == 1 && c20 == 1) {                           // :493
        const float th = (float)acos(c12 == 1 ? t1a : t2a);            // one acos; (float)(-x) == -(float)x
        if (c12 == 1) { r.sel = 1; r.dir = th; }                       // :498-501
        else { r.sel = 2; r.dir = -th; }                               // :507-510
        return r;
      }
      fallback = (c12 == 1) ? 1 : 2;                        // :518 / :536
    } else {
      const double theta0 = acos(t0a), theta1 = acos(t1a), theta2 = acos(t2a);  // :456-458
      if (theta1 < theta0 && theta2 < theta0) {             // :493
        if (theta1 < theta2) { r.sel = 1; r.dir = (float)theta1; }     // :498-501
        else { r.sel = 2; r.dir = (float)(-theta2); }                  // :507-510
        return r;
      }
      fallback = (theta1 < theta2) ? 1 : 2;                 // :518 / :536
    }
  }
  r.kind = 2; r.sel = fallback;
  r.cand = (fallback == 1) ? (u1 + b) : (u2 + a);           // :420,439,520,538
  return r;
}

// Gates of :411 and :421,440,521,539 against the current value; returns the reference's bool.
MNAV_HD bool cvp_apply(const CvpCand& k, float& u3, int& sel, float& dir)
{
  if (!(k.u3tmp < (double)u3)) return false;                // :411
  if (k.kind == 2 && !(k.cand < (double)u3)) return false;  // :421,440,521,539
  u3 = (float)k.cand; sel = k.sel; dir = k.dir;             // :430,449,497,526,544
  return true;
}

MNAV_HD CvpUpd cvp_update(float u1f, float u2f_, float u3f, float af, float bf, float cf)
{
  CvpUpd r; r.u3 = u3f; r.dir = 0.0f; r.sel = 0;
  const CvpCand k = cvp_candidate(u1f, u2f_, af, bf, cf);
  r.ok = cvp_apply(k, r.u3, r.sel, r.dir);
  return r;
}

// ---------------------------------------------------------------------------------------
// Inflation triangle update, inflation_layer.cpp:181-313, on plain numbers: float32 throughout (the reference's
// locals are float and <math.h> supplies the float overloads of sqrt), no FMA contraction.
// ---------------------------------------------------------------------------------------
constexpr float kLayersEpsilon = 1e-9f;                    // mesh_layers::EPSILON, inflation_layer.h:48

// computeUpdateSethianMethod(d1, d2, a, b, dot, F = 1.0) :181-234
MNAV_HD float infl_sethian(float d1, float d2, float a, float b, float dot)
{
  const float F = 1.0f;
  float t = inf_f();                                        // :186
  const float rc = dot;                                     // :188
  const float rs = sqrtf(1 - dot * dot);                    // :189
  const float u = d2 - d1;                                  // :191
  const float f2 = a * a + b * b - 2 * a * b * rc;          // :193
  const float f1 = b * u * (a * rc - b);                    // :194
  const float f0 = b * b * (u * u - F * F * a * a * rs);    // :195
  const float delta = f1 * f1 - f0 * f2;                    // :197
  if (delta >= 0) {                                         // :199
    if (fabsf(f2) > kLayersEpsilon) {                       // :201
      t = (-f1 - sqrtf(delta)) / f2;                        // :203
      if (t < u || b * (t - u) / t < a * rc || a / rc < b * (t - u) / 2) {   // :204
        t = (-f1 + sqrtf(delta)) / f2;                      // :206
      } else {
        if (f1 != 0) t = -f0 / f1;                          // :210-213
        else t = -inf_f();                                  // :216
      }
    }
  } else {
    t = -inf_f();                                           // :223
  }
  if (u < t && a * rc < b * (t - u) / t && b * (t - u) / t < a / rc) return t + d1;   // :226-229
  return fminf(b * F + d1, a * F + d2);                     // :232
}

// waveFrontUpdate :236-313 without its stores: the value offered to the free vertex and whether a successful
// update re-queues it (:311).  ok == false: u3tmp is not finite (:271).
struct InflCand { float u3tmp; bool ok; bool requeue; };
MNAV_HD InflCand infl_candidate(float u1, float u2, float a, float b, float c, float max_distance)
{
  InflCand r;
  const float dot = (a * a + b * b - c * c) / (2 * a * b);  // :260-268
  r.u3tmp = infl_sethian(u1, u2, a, b, dot);                // :269
  r.ok = (f2u(r.u3tmp) & 0x7f800000u) != 0x7f800000u;       // std::isfinite :271
  r.requeue = u1 <= max_distance && u2 <= max_distance;     // :311
  return r;
}

// ---------------------------------------------------------------------------------------
// Per-plan control block.  Two copies ping-pong between consecutive steps (step j reads
// slot (j-1)&1 and block 0 writes slot j&1); three counter blocks rotate (step j counts
// into j%3, reads (j-1)%3, clears (j+1)%3).  See DESIGN.md §3.3.
// ---------------------------------------------------------------------------------------
struct Ctl {
  int32_t it;          // index of the step that produced this block (-1 = initial state)
  uint32_t n;          // entries in the work list that step `it` processes
  float thr;           // band upper bound: pop times < thr are settled by this band
  float thr_fixed;     // pop times < thr_fixed are final (bands completed so far)
  float goal_dist;     // +inf until armed (dijkstra :296 / cvp :769)
  uint32_t armed;
  uint32_t done;
  uint32_t band_new;   // 1 on the first step of a band
  uint32_t bands;      // statistics
  uint32_t overflow;   // work-list overflow (never with capacity V; reported as internal error)
  uint32_t repair;     // 1: post-arming repair sweep over all vertices, 2: work-list rebuild after a band shrink
  uint32_t evals;      // statistics: vertex evaluations so far
  uint32_t band_steps; // steps spent in the current band
  float width;         // current band width (<= Plan.delta; shrinks when a band does not converge)
  uint32_t shrinks;    // statistics: band shrinks
  // waiting list (see Plan.wlist): keyed vertices beyond the band wait there instead of being carried through every step
  uint32_t wsel;       // buffer the current band appends to
  uint32_t wread;      // epoch steps only: entries of the OTHER buffer (the previous band's waiting list) to process
  uint32_t wbase;      // entries already in the current buffer when this step starts (0 in an epoch step)
  float wmin;          // smallest pop time parked in the current buffer before this step (+inf in an epoch step); may be
                       // stale-low when a parked vertex moved up later: the next band then starts lower, never wrong
  uint32_t cuts;       // statistics: band cuts (kCutAfter)
  uint32_t epoch;      // id of the current waiting list (1 for the list the first steps append to, then step index of the
                       // epoch step + 2); Plan.wstamp (0 = never parked) dedups with it
  uint32_t serial;     // (unused since round 6: the exact band routine replaced the serial band)
  uint32_t arm_vertex; // CVP: the robot-face vertex whose pop armed goal_dist (kNone until then): pops up to and including its
                       // own met goal_dist = +inf at :754 and expand whatever their value (passes_goal_cut)
  // the exact band (exact_band_* below): a band that does not settle is handed to a routine that pops its vertices one at a time
  uint32_t exact_wanted; // 1: the steps idle (repair == 5) until the host has run the exact band routine on this plan
  uint32_t force_cut;    // 1: the next step is a cut step at thr (rebuilds the waiting list), then the band starts again
  uint32_t exact;        // inside the routine only: a support is fixed when it was settled by an earlier band or popped by the routine ...
  uint32_t bound_v;      // ... i.e. when its key is not after the key of this vertex, the last one the routine popped (kNone: none yet)
};

struct Cnt {
  uint32_t n_next;     // entries pushed to the next work list
  uint32_t changed;    // in-band vertices whose (dist, pop time) changed this step
  uint32_t minkey;     // min pop time (float bits) over the entries parked this step
  uint32_t evals;      // statistics: vertex evaluations
  uint32_t n_wait;     // entries appended to the waiting list this step
  uint32_t minchg;     // min pop time (float bits) over the vertices that moved ACROSS the band's upper bound this step
                       // (in band before and beyond it now, or the other way round; see kCutAfter)
  uint32_t pad[2];
};

// A pointer field of a plan record.  In the device pass element access goes through the GLOBAL address space: a pointer loaded
// from a record is otherwise a generic one and every access through it a flat_load, which counts on both memory counters and is
// waited for with vmcnt(0) lgkmcnt(0) -- no two dependent-free loads ever overlap (k_step_wide had 323 of them).  Stored and
// converted as a plain pointer (same layout in both passes, the host fills in device pointers; `P.dist + v`, atomics, null tests
// see a T*); only p[i], *p and p-> are typed global.
#if defined(__HIP_DEVICE_COMPILE__)
#define MNAV_GP __attribute__((address_space(1)))
#else
#define MNAV_GP
#endif
template <class T> struct GPtr {
  T* p;
  GPtr() = default;
  MNAV_HD GPtr(T* q) : p(q) {}
  MNAV_HD operator T*() const { return p; }
  template <class I> MNAV_HD T MNAV_GP& operator[](I i) const { return ((T MNAV_GP*)p)[i]; }
  MNAV_HD T MNAV_GP& operator*() const { return *(T MNAV_GP*)p; }
  MNAV_HD T MNAV_GP* operator->() const { return (T MNAV_GP*)p; }
};

// Constant per-plan parameters + state pointers.  All pointers address the plan's own slices.
struct Plan {
  uint32_t planner;        // kPlannerDijkstra / kPlannerCvp
  uint32_t V;
  // shared read-only mesh data
  GPtr<const uint32_t> row_ptr; // V+1 (Dijkstra gather CSR)
  GPtr<const Nbr> nbr;          // 2E
  GPtr<const uint32_t> crn_ptr; // V+1 (CVP corners)
  GPtr<const Corner> crn;       // 3F
  GPtr<const uint8_t> blocked;  // V: CVP free-vertex gate (cost >= limit || invalid), cvp :760,802,825,848
  // per-plan state
  GPtr<float> dist;             // V  potential
  GPtr<PopKey> tkey;            // V  pop key (CVP only)
  GPtr<uint32_t> pred;          // V
  GPtr<float> dirn;             // V  (CVP)
  GPtr<uint32_t> cutf;          // V  (CVP)
  GPtr<uint32_t> stamp;         // V  work-list dedup
  GPtr<uint32_t> dirty;         // V  step for which a neighbour asked for a re-evaluation
  GPtr<uint32_t> list[2];       // work lists, capacity `cap`: vertices to (re-)evaluate in the next step
  GPtr<uint32_t> wlist[2];      // waiting lists, capacity `cap`: keyed vertices beyond the band.  They are looked at again when
                           // a neighbour moves (which puts them on the work list) or when the band advances (the epoch
                           // step of the next band processes the whole list), not in every step in between
  GPtr<uint32_t> wstamp;        // V  epoch in which the vertex was last appended to a waiting list
  uint32_t cap;
  GPtr<Ctl> ctl;                // [2]
  GPtr<Cnt> cnt;                // [4]: three rotating step counters + cnt[3] = sticky flags of the plan (kFlag*)
  // parameters
  float delta;             // band width
  double offset;           // goal_dist_offset
  uint32_t goal_tie1;      // 1 + the id that stands for the robot vertex in (value, id) ties of goal_cut (a part of a partitioned mesh that
                           // does not hold the robot vertex: its rank among the part's ids); 0: the robot vertex itself
  uint32_t seed[3];        // wave seed vertices (Dijkstra: seed[0], others kNone)
  uint32_t seed_expands[3];// seed passes the cost/invalid cut-offs (cvp :757,760)
  float seed_d[3];         // initial potential of the seeds (Dijkstra 0; CVP Euclidean, cvp :721-723)
  uint32_t seed_face;      // CVP: cutting face of the seeds (cvp :725)
  uint32_t target[3];      // robot vertex / robot-face vertices
  uint32_t target_expands[3];
  uint32_t max_steps;
  int32_t walk_max;        // bound of key_less / key_for walks (kKeyWalkMax)
  int32_t descend_max;     // bound of key_descends_from (kDescendWalkMax)
  // Inflation wave (InflationLayer::waveCostInflation, inflation_layer.cpp:341-491) on the CVP machinery: set iff
  // seed_mask != nullptr.  seed_mask[v]: kInflSeed = lethal vertex (distance 0, fixed from the start :397-402),
  // kInflSeedMute = lethal and invalid (fixed, but its pop is skipped :417), kInflMute = invalid free vertex (is
  // updated, queued and popped like any other, but the pop is skipped before `fixed` is set :417-422: never a support).
  GPtr<const uint8_t> seed_mask;
  GPtr<float> keyd;             // V  value the vertex sits in the queue with (the last update that re-queued it, :311,:451)
  float infl_max;          // max_distance: an update only (re-)queues its vertex while both supports lie within (:311)
};
static_assert(sizeof(GPtr<float>) == sizeof(float*) && alignof(GPtr<float>) == alignof(float*), "GPtr: a pointer, nothing else (the host fills plan records with memcpy)");
static_assert(__is_trivially_copyable(Plan) && __is_trivially_copyable(GPtr<float>), "plan records travel by hipMemcpy");
enum : uint8_t { kInflFree = 0, kInflSeed = 1, kInflSeedMute = 2, kInflMute = 3 };

// Sticky per-plan flags (P.cnt[3].n_next).  kFlagWalkLimit: a walk over the cascade tree hit its bound and
// the comparison fell back to another order -- the result may no longer be the reference's, so the plan is
// reported as INTERNAL_ERROR instead of being returned (racy stores of the same bit: benign).
constexpr uint32_t kFlagWalkLimit = 1u;
MNAV_HD void raise_flag(const Plan& P, uint32_t f) { if (!(P.cnt[3].n_next & f)) P.cnt[3].n_next |= f; }

MNAV_HD bool is_seed(const Plan& P, uint32_t v)
{
  if (P.seed_mask) { const uint8_t m = P.seed_mask[v]; return m == kInflSeed || m == kInflSeedMute; }
  return v == P.seed[0] || v == P.seed[1] || v == P.seed[2];
}

// --- pop keys as positions in the cascade forest (see PopKey) --------------------------------
// (inflation: a vertex can sit in the queue with an older, larger value than its distance -- keyd, see Plan)
MNAV_HD KeyRef key_ref(const Plan& P, uint32_t u) { return key_ref_of(P.tkey[u], P.keyd ? P.keyd[u] : P.dist[u], u); }

// Walks over the cascade tree follow `up` links.  A half-converged tree can hold anything, even CYCLES of links, and a
// walk that runs around one until the bound costs milliseconds (one dependent global load per step).  CycleGuard is
// Brent's detector: it remembers a node of the walk, at doubling distances, and reports when the walk comes back to it.
struct CycleGuard {
  unsigned long long mark; int span, n;
  MNAV_HD void start(unsigned long long own) { mark = own; span = 1; n = 0; }
  MNAV_HD bool visit(unsigned long long own)                       // true: this node was seen before
  {
    if (own == mark) return true;
    if (++n == span) { mark = own; span <<= 1; n = 0; }
    return false;
  }
};

// a pops before b, for two nodes of the SAME cascade (equal `hi`): the walk over the tree.  Rare against the one-compare
// case of key_less below, and heavy on registers: a real call, not inlined into the replay.
MNAV_HD_COLD bool key_less_walk(const Plan& P, KeyRef a, KeyRef b)
{
  int guard = 0;
  CycleGuard ga, gb;
  ga.start(a.own); gb.start(b.own);
  for (; guard < P.walk_max; ++guard) {                            // same cascade: preorder, siblings by (value, id)
    if (a.k.lvl == b.k.lvl) {
      if (a.own == b.own) return false;                            // the same node
      if (a.k.lvl == 0u || a.k.up == b.k.up) return a.own < b.own; // siblings
      a = key_ref(P, a.k.up); b = key_ref(P, b.k.up);
      if (ga.visit(a.own) || gb.visit(b.own)) break;
    } else if (a.k.lvl > b.k.lvl) {
      if (a.k.up == pair_id(b.own)) return false;                  // b is an ancestor of a: pops first
      if (a.k.up == kNone) break;
      a = key_ref(P, a.k.up);
      if (ga.visit(a.own)) break;
    } else {
      if (b.k.up == pair_id(a.own)) return true;
      if (b.k.up == kNone) break;
      b = key_ref(P, b.k.up);
      if (gb.visit(b.own)) break;
    }
  }
  // Left the loop without a decision: the two nodes are not in one consistent tree.  In a half-converged state
  // that is transient and any total order will do (the vertex is evaluated again); a walk that ran into its
  // bound on a CONVERGED tree would silently change the order, so it is flagged.
  if (guard == P.walk_max) raise_flag(P, kFlagWalkLimit);
  return a.own < b.own;
}

// a pops before b
MNAV_HD bool key_less(const Plan& P, const KeyRef& a, const KeyRef& b)
{
  if (a.k.hi != b.k.hi) return a.k.hi < b.k.hi;                    // different main-front pops
  if (a.k.lvl == b.k.lvl && a.own == b.own) return false;          // the same node (key_less_walk's first test, without the call:
  return key_less_walk(P, a, b);                                   //  two faces fired by one pop are compared all the time)
}

// cvp :754 `if (distances[cur] > goal_dist) continue;` as a test on the converged state.  goal_dist is +inf until the arming
// pop (:765-769, which comes AFTER :754 in the same iteration), so every pop up to and including that one expands whatever
// its value; later pops expand when their value does not exceed goal_dist.  With goal_dist_offset >= 0 the first clause
// decides nearly always; a negative offset (goal_dist below the arming vertex's own value) is where the pop order does.
MNAV_HD_COLD bool popped_by_arming(const Plan& P, uint32_t arm_vertex, KeyRef k)   // rare (values above goal_dist only): a real call
{
  return !key_less(P, key_ref(P, arm_vertex), k);                    // popped no later than the arming vertex
}
MNAV_HD bool passes_goal_cut(const Plan& P, const Ctl& c, float d, const KeyRef& k)
{
  if (!(d > c.goal_dist)) return true;
  if (c.arm_vertex == kNone) return false;
  return popped_by_arming(P, c.arm_vertex, k);
}

// key of vertex v whose value d was set by the pop `trig`
MNAV_HD PopKey key_for(const Plan& P, float d, uint32_t v, KeyRef trig)
{
  const unsigned long long x = key_pair(d, v);
  PopKey k;
  if (x > trig.k.hi) { k.hi = x; k.up = kNone; k.lvl = 0u; return k; }   // at or above the main front: ordinary pop
  k.hi = trig.k.hi;                                                // below it: inside the cascade of trig's root
  KeyRef a = trig;                                                 // climb to the node whose sub-cascade v pops in:
  int guard = 0;
  CycleGuard ga;
  ga.start(a.own);
  for (; guard < P.walk_max; ++guard) {                            // the deepest ancestor-or-self of trig above v
    if (a.k.lvl == 0u || x < a.own || a.k.up == kNone) break;
    a = key_ref(P, a.k.up);
    if (ga.visit(a.own)) break;
  }
  if (guard == P.walk_max) raise_flag(P, kFlagWalkLimit);
  k.up = pair_id(a.own); k.lvl = a.k.lvl + 1u;
  return k;
}

// Does the stored key of t place it inside the sub-cascade of v (v is an ancestor of t)?  Such a t pops
// after v, so it can never be a trigger FOR v; in a half-converged state it must not act as one either,
// or the two would keep supporting each other (v set by its own child, the child by v, ...).
constexpr int kDescendWalkMax = 64;
MNAV_HD bool key_descends_from_pre(const Plan& P, PopKey a, uint32_t v)   // a = tkey[t]
{
  int guard = 0;
  for (; guard < P.descend_max && a.lvl > 0u; ++guard) {
    if (a.up == v) return true;
    if (a.up == kNone) return false;
    a = P.tkey[a.up];
  }
  if (guard == P.descend_max && a.lvl > 0u) raise_flag(P, kFlagWalkLimit);
  return false;
}
MNAV_HD bool key_descends_from(const Plan& P, uint32_t t, uint32_t v) { return key_descends_from_pre(P, P.tkey[t], v); }   // (no shortcut through hi / lvl: both may be stale)

// --- which sources the Dijkstra wave expands, as a test on FINAL values ------------------------
// The reference pops vertices in (value, id) order, arms goal_dist = float(dist[target] + offset) when the robot vertex pops
// (dijkstra :293-297) and from then on skips every popped vertex above it (:299).  Everything popped BEFORE the robot vertex
// was expanded whatever the offset.  So with goal_dist >= dist[target] the expanded set is {d <= goal_dist}; with a negative
// offset that rounds below dist[target] it is {popped before the robot vertex} = {d < dt or (d == dt and id < target)}, the
// robot vertex itself not among them.  One predicate for both: d < cut || (d == cut && id < tie).
struct GoalCut { float goal, cut; uint32_t tie; };
MNAV_HD GoalCut goal_cut(float dt, double offset, uint32_t target)
{
  GoalCut g; g.goal = inf_f(); g.cut = inf_f(); g.tie = kNone;
  if (dt < inf_f()) {
    g.goal = (float)((double)dt + offset);                             // dijkstra :296
    g.cut = g.goal;
    if (g.goal < dt) { g.cut = dt; g.tie = target; }
  }
  return g;
}
MNAV_HD bool expanded_source(const GoalCut& g, float d, uint32_t id) { return d < inf_f() && (d < g.cut || (d == g.cut && id < g.tie)); }

// --- arming of goal_dist once the robot vertex / robot face is settled -----------------------
// Dijkstra: goal_dist = dist[target] + offset when the target pops (dijkstra :293-297).
// CVP: when a robot-face vertex pops that passes the cut-offs while all three are fixed
// (cvp :757-771); seeds are fixed from the start (cvp :726).
MNAV_HD void try_arm(const Plan& P, Ctl& q)
{
  if (P.planner == kPlannerDijkstra) {
    const uint32_t t = P.target[0];
    if (t == kNone) return;
    const float d = P.dist[t];
    if (d < q.thr_fixed) { q.goal_dist = (float)((double)d + P.offset); q.armed = 1; }
    return;
  }
  KeyRef k_all = key_ref_of(key_inf(), 0.0f, 0); bool have_all = false;   // pop key of the last goal vertex to get fixed
  for (int k = 0; k < 3; ++k) {
    const uint32_t g = P.target[k];
    if (g == kNone) return;
    if (is_seed(P, g)) continue;                 // fixed from the start
    const KeyRef kg = key_ref(P, g);
    if (!(key_time(kg.k) < q.thr_fixed)) return; // not all fixed yet
    if (!have_all || key_less(P, k_all, kg)) { k_all = kg; have_all = true; }
  }
  KeyRef best_k = k_all; float best_d = 0.0f; uint32_t best_i = kNone;
  for (int k = 0; k < 3; ++k) {
    const uint32_t g = P.target[k];
    if (!P.target_expands[k]) continue;
    const KeyRef kg = key_ref(P, g);
    if (!(key_time(kg.k) < q.thr_fixed)) continue; // has not popped yet
    if ((!have_all || !key_less(P, kg, k_all)) && (best_i == kNone || key_less(P, kg, best_k))) { best_k = kg; best_d = P.dist[g]; best_i = g; }
  }
  if (best_i != kNone) { q.goal_dist = (float)((double)best_d + P.offset); q.armed = 1; q.arm_vertex = best_i; }
}

// Band controller: pure function of the previous control block and the previous step's counters.
//
// goal_dist is only known once the robot vertex/face is settled, and the band that settles it
// may already have let vertices beyond goal_dist act as sources.  Everything at or below
// goal_dist is unaffected by that (sources only feed larger values), so arming is followed by
// one REPAIR step that re-evaluates every vertex above goal_dist under the final cut-off and
// rebuilds the work list (process_repair below).
constexpr uint32_t kBandStepLimit = 64;   // a band that is still moving after this many steps is cut down
// Before that, the cheap remedy.  What keeps a band from settling is almost always a small cluster of cascade members
// whose states straddle the band's upper bound: in one state a vertex pops just below `thr` (and supports its
// neighbours), in the other just above (and does not), and the cluster flips between the two.  After kCutAfter steps
// the band is CUT right below the lowest vertex that still moved across the bound (a band that merely needs many
// steps -- a deep cascade unwinding -- has no such vertex and is left alone): everything below is converged and settles, the
// cluster falls entirely into the next band, where both of its states are in band.  The cut costs one scan step
// (repair == 3: no evaluation, keyed vertices at or above the cut are parked, the work list is carried over).
// Used by the inflation wave only (one band per radius, the whole wave sits at the bound); the CVP planner's bands are
// narrow against its wave and a cut costs it more steps than the rare kBandStepLimit shrink (measured: -15 % plans/s).
constexpr uint32_t kCutAfter = 24, kCutMaxList = 512;   // ... and only when the work list has shrunk to a remainder (a band
                                                         // that is still making progress has thousands of entries)

MNAV_HD Ctl controller_core(const Plan& P, const Ctl& p, const Cnt& c, float m_wait, uint32_t n_wait)
{
  Ctl q = p;
  q.it = p.it + 1;
  q.repair = 0;
  q.evals = p.evals + c.evals;
  if (p.done) { q.n = 0; return q; }
  q.n = c.n_next;
  if (c.n_next > P.cap) { q.overflow = 1; q.done = 1; q.n = 0; return q; }
  const bool out_of_steps = (uint32_t)q.it >= P.max_steps;
  if (p.exact_wanted && !out_of_steps) { q.repair = 5; q.n = 0; return q; }   // idle until the exact band routine has run
  if (p.force_cut && !out_of_steps) {                                 // the exact band routine has run: rebuild the waiting list, start the band again
    q.force_cut = 0; q.repair = 3; q.band_steps = 0; q.n = 0;
    return q;
  }
  if (p.repair == 3 && !out_of_steps) {                               // the band was cut: start it again under the lower bound
    q.band_new = 1; q.band_steps = 0;
    return q;
  }
  if (p.repair == 1 && c.changed > 0 && !out_of_steps) {             // repair sweep not yet at its fixed point (CVP)
    q.repair = 1; q.band_new = 0;
    return q;
  }
  if (c.changed > 0 && !out_of_steps) {
    q.band_new = 0;
    q.band_steps = p.band_steps + 1;
    if (P.seed_mask != nullptr && q.band_steps >= kCutAfter && q.band_steps < kBandStepLimit && !p.repair && !p.band_new &&
        c.n_next <= kCutMaxList) {
      const float lo = p.thr_fixed > 0.0f ? p.thr_fixed : 0.0f;
      const float cut = u2f(c.minchg);
      if (cut > next_up(lo) && cut < p.thr) {                          // (otherwise: keep stepping, kBandStepLimit is the backstop)
        q.thr = cut; q.repair = 3; q.band_steps = 0; q.cuts = p.cuts + 1;
        return q;
      }
    }
    const bool wide = p.thr > next_up(p.thr_fixed > 0.0f ? p.thr_fixed : 0.0f);
    // (inflation waves: after three shrinks in a row -- a sixty-fourth of the radius -- the band goes to the exact band routine
    //  as it is; shrinking on, down to single keys, cost 66 steps per key on maps full of tied pop times: step cap)
    const bool shrink = wide && (P.seed_mask == nullptr || p.width > P.delta * (1.0f / 64.0f));
    if (P.planner == kPlannerCvp && q.band_steps >= kBandStepLimit && shrink) {
      // Not converging: on triangles that grossly violate the triangle inequality the in-band
      // vertices can support each other in a cycle.  Cut the band down from the bottom (at the
      // width of a single key the replay is exactly the sequential loop) and rebuild the work
      // list with a full scan; the width recovers over the following bands.
      const float lo = p.thr_fixed > 0.0f ? p.thr_fixed : 0.0f;
      q.width = p.width * 0.25f;
      float thr = lo + q.width;
      if (!(thr > lo)) thr = next_up(lo);
      if (thr > p.thr) thr = p.thr;
      q.thr = thr;
      q.repair = 2; q.band_new = 1; q.band_steps = 0; q.shrinks = p.shrinks + 1;
    } else if (P.seed_mask != nullptr && q.band_steps >= kBandStepLimit) {
      // A narrow band that is STILL moving: vertices of exactly the same pop time (ties are common around isolated lethal
      // vertices on a regular grid) and the cascades below them, whose members support each other with PROVISIONAL keys and keep
      // re-hanging each other -- under the device's concurrent in-place evaluation, under a sequential pass that inherits its
      // state, and (2-3 % -> 0.5 % of random sparse-lethal maps, round 6) even under a sequential pass from a clean state;
      // reproduced on the CPU model with a seeded mixture of snapshot and in-place reads (orders >= 4).  What cannot cycle is the
      // reference's own procedure -- one pop at a time, a vertex acting as a support only once its state is final --: the band is
      // handed to exact_band_* (the steps idle until the host has run it).  (Rounds 5-6 first ran such a band entry after entry on
      // one 8-lane group, `serial`, then the same from a reset state: gone, the routine covers both.)
      q.exact_wanted = 1; q.repair = 5; q.n = 0;
    }
    return q;                                                          // band still moving
  }
  q.thr_fixed = p.thr;
  q.band_steps = 0;
  if (!p.repair) q.bands = p.bands + 1;
  if (!q.armed && !out_of_steps) {
    try_arm(P, q);
    if (q.armed) { q.repair = 1; q.band_new = 0; return q; }
  }
  const float m = m_wait;                                              // smallest pop time on the waiting list
  if (out_of_steps) { q.overflow = 2; q.done = 1; q.n = 0; return q; }   // did not converge: reported as an error
  if ((c.n_next == 0 && n_wait == 0) || !(m < inf_f())) { q.done = 1; q.n = 0; return q; }
  if (q.armed && m > q.goal_dist) { q.done = 1; q.n = 0; return q; }   // nothing left that may expand
  q.width = fminf(P.delta, fmaxf(p.width * 2.0f, P.delta * (1.0f / 4096.0f)));   // (recovers from any shrink: a width that underflowed stayed 0)
  float thr = m + q.width;
  if (!(thr > m)) thr = next_up(m);
  q.thr = thr;
  q.band_new = 1;
  return q;
}

// The waiting-list bookkeeping around controller_core: totals of the list the previous step appended to, and the
// switch to the other buffer whenever the next step starts a new epoch (first step of a band, repair / rebuild sweeps).
MNAV_HD Ctl controller(const Plan& P, const Ctl& p, const Cnt& c)
{
  const uint32_t wtot = p.wbase + c.n_wait;
  const float wmin = fminf(p.wmin, u2f(c.minkey));
  Ctl q = controller_core(P, p, c, wmin, wtot);
  q.wsel = p.wsel; q.wread = 0; q.wbase = wtot; q.wmin = wmin; q.epoch = p.epoch;
  if (!q.done && (q.band_new || (q.repair && q.repair != 5))) {
    q.wsel = p.wsel ^ 1u; q.wread = q.repair ? 0u : wtot; q.wbase = 0; q.wmin = inf_f(); q.epoch = (uint32_t)q.it + 2u;
  }
  return q;
}

// ---------------------------------------------------------------------------------------
// Gather rules
// ---------------------------------------------------------------------------------------
struct Eval { float d; float t; PopKey key; uint32_t pred; float dir; uint32_t cut; float keyd; };

// Dijkstra: dist[v] = min over neighbours u that expand (popped: dist[u] < thr; not cut off:
// dist[u] <= goal_dist, dijkstra :299; cost cut-off folded into w) of dist[u] + w(u,v), the very
// float add of dijkstra :331.  Predecessor = first-popped neighbour attaining the minimum
// (strict '<' at :332): argmin (sum, dist[u], u) -- DESIGN.md "tie rule".
MNAV_HD Eval eval_dijkstra(const Plan& P, const Ctl& c, uint32_t v)
{
  Eval e; e.d = inf_f(); e.pred = v; e.dir = 0.0f; e.cut = kNone; e.key = key_inf();
  float best_du = inf_f();
  const uint32_t beg = P.row_ptr[v], end = P.row_ptr[v + 1];
  for (uint32_t i = beg; i < end; ++i) {
    const Nbr nb = P.nbr[i];
    const float du = P.dist[nb.u];
    if (!(du < c.thr) || du > c.goal_dist) continue;
    const float s = du + nb.w;
    if (s < e.d || (s == e.d && s < inf_f() && (du < best_du || (du == best_du && nb.u < e.pred)))) {
      e.d = s; best_du = du; e.pred = nb.u;
    }
  }
  if (!(e.d < inf_f())) e.pred = v;
  e.t = e.d;
  return e;
}

// Fire event of face (v1,v2 -> v): the pop of a support that passes the cut-offs (cvp :754-760)
// while the other support is already fixed (seed, or popped earlier).  Returns the pop key of the
// earliest such pop; trig == kNone when the face cannot fire in this band.
struct Fire { KeyRef key; uint32_t trig; };

// Has the support with pop time t / key k popped as far as this evaluation is concerned?  In a band step: everything below the
// band's upper bound (the in-band vertices support each other with their provisional keys, the iteration sorts it out).  In the
// exact band routine: what an earlier band settled, and what the routine has popped so far.
MNAV_HD_COLD bool popped_exact(const Plan& P, const Ctl& c, float t, KeyRef k)
{
  if (t < c.thr_fixed) return true;
  if (c.bound_v == kNone || !(t < inf_f())) return false;
  return !key_less(P, key_ref(P, c.bound_v), k);                     // not after the last pop
}
MNAV_HD bool popped(const Plan& P, const Ctl& c, float t, const KeyRef& k)
{
  if (!c.exact) return t < c.thr;
  return popped_exact(P, c, t, k);
}

// (k1, k2: key_ref of the two supports, d1, d2: their potentials -- loaded by the caller, so that a kernel can have the loads of
// many faces in flight before the first one is looked at)
MNAV_HD Fire corner_fire_pre(const Plan& P, const Ctl& c, const Corner& k, const KeyRef& k1, const KeyRef& k2, float d1, float d2)
{
  Fire f; f.key = key_ref_of(key_inf(), inf_f(), 0); f.trig = kNone;
  if (k.v1 == kNone) return f;
  const bool s1 = is_seed(P, k.v1), s2 = is_seed(P, k.v2);
  const float t1 = key_time(k1.k), t2 = key_time(k2.k);
  const bool p1 = popped(P, c, t1, k1), p2 = popped(P, c, t2, k2);
  if (!((s1 || p1) && (s2 || p2))) return f;                         // both supports fixed by this band
  bool ex1 = true, ex2 = true;
  if (P.seed_mask) {
    const uint8_t m1 = P.seed_mask[k.v1], m2 = P.seed_mask[k.v2];
    if (m1 == kInflMute || m2 == kInflMute) return f;                 // never fixed: the face never has two fixed supports
    ex1 = m1 != kInflSeedMute; ex2 = m2 != kInflSeedMute;
  } else {
    if (s1) { for (int q = 0; q < 3; ++q) if (P.seed[q] == k.v1) ex1 = P.seed_expands[q] != 0; }
    if (s2) { for (int q = 0; q < 3; ++q) if (P.seed[q] == k.v2) ex2 = P.seed_expands[q] != 0; }
  }
  const bool one_first = key_less(P, k1, k2);                        // v1 pops before v2
  const bool trig1 = p1 && ex1 && passes_goal_cut(P, c, d1, k1) && (s2 || !one_first);
  const bool trig2 = p2 && ex2 && passes_goal_cut(P, c, d2, k2) && (s1 || one_first || k1.own == k2.own);
  if (trig1) { f.key = k1; f.trig = k.v1; }
  if (trig2 && (!trig1 || key_less(P, k2, k1))) { f.key = k2; f.trig = k.v2; }
  return f;
}

MNAV_HD Fire corner_fire(const Plan& P, const Ctl& c, const Corner& k)
{
  if (k.v1 == kNone) { Fire f; f.key = key_ref_of(key_inf(), inf_f(), 0); f.trig = kNone; return f; }
  return corner_fire_pre(P, c, k, key_ref(P, k.v1), key_ref(P, k.v2), P.dist[k.v1], P.dist[k.v2]);
}

// The SECOND fire event of a face, where it has one.  A face is visited at the pop of a support whenever its other support
// is fixed then (cvp :790-870).  A non-seed support is fixed by its own pop, so such a face fires once, at the later pop; a SEED is
// fixed from the start (:726) but pops like everybody else -- a face with a seed support fires at the pop of its other support
// AND, when that one comes first, again at the seed's own pop.  The second visit offers the very candidate of the first; in exact
// arithmetic it could never lower anything, but :411 compares the float64 candidate with the float32 value that was stored --
// usually a hair above it -- so the re-application "succeeds", leaves the value as it is and makes this face the vertex's cutting
// face (and its support the predecessor) again where a tying face had taken them in between.  The replay of the step kernels uses
// the first event only (corner_fire_pre); seed_ring_fix replays both for the few vertices around the seed face.
MNAV_HD Fire corner_fire_second(const Plan& P, const Ctl& c, const Corner& k)
{
  Fire f; f.key = key_ref_of(key_inf(), inf_f(), 0); f.trig = kNone;
  if (k.v1 == kNone || P.seed_mask) return f;
  const bool s1 = is_seed(P, k.v1), s2 = is_seed(P, k.v2);
  if (!s1 && !s2) return f;
  const KeyRef k1 = key_ref(P, k.v1), k2 = key_ref(P, k.v2);
  const float t1 = key_time(k1.k), t2 = key_time(k2.k);
  if (!((s1 || t1 < c.thr) && (s2 || t2 < c.thr)) || k1.own == k2.own) return f;
  bool ex1 = true, ex2 = true;
  if (s1) { for (int q = 0; q < 3; ++q) if (P.seed[q] == k.v1) ex1 = P.seed_expands[q] != 0; }
  if (s2) { for (int q = 0; q < 3; ++q) if (P.seed[q] == k.v2) ex2 = P.seed_expands[q] != 0; }
  const bool one_first = key_less(P, k1, k2);
  const bool trig1 = t1 < c.thr && ex1 && passes_goal_cut(P, c, P.dist[k.v1], k1) && (s2 || !one_first);
  const bool trig2 = t2 < c.thr && ex2 && passes_goal_cut(P, c, P.dist[k.v2], k2) && (s1 || one_first);
  if (!(trig1 && trig2)) return f;
  if (one_first) { f.key = k2; f.trig = k.v2; } else { f.key = k1; f.trig = k.v1; }     // the later of the two pops
  return f;
}

// CVP: replay of the incident-face updates of vertex v in the order their trigger vertices
// pop.  A face is applied to v only while v is still free, i.e. while the trigger pops before v
// itself would.  Faces fired by the same pop are applied in the order of the trigger's half-edge
// circulator (cvp :778 loop over getFacesOfVertex(trigger)): Corner order flags, mnav_build.h.
template <bool BOTH_EVENTS>
MNAV_HD Eval eval_cvp_t(const Plan& P, const Ctl& c, uint32_t v)
{
  Eval e; e.d = inf_f(); e.t = inf_f(); e.key = key_inf(); e.pred = v; e.dir = 0.0f; e.cut = kNone; e.keyd = inf_f();
  const uint32_t beg = P.crn_ptr[v], end = P.crn_ptr[v + 1];
  const bool infl = P.seed_mask != nullptr;
  // an invalid free vertex is queued and popped like the others, but its pop is skipped before it is fixed (:417-422):
  // it never stops taking updates and never supports one -> it simply has no pop key here
  const bool mute = infl && P.seed_mask[v] == kInflMute;
  KeyRef last = key_ref_of(key_inf(), inf_f(), 0);
  bool first = true, queued = false;
  // every pass consumes one trigger vertex (<= 2 per corner); on a half-converged, inconsistent cascade tree the
  // "next pop after the last one" can run in a circle -- transient there, flagged if it happens on a converged state
  const uint32_t max_pass = 2u * (end - beg) + 2u;
  for (uint32_t pass_no = 0;; ++pass_no) {
    if (pass_no == max_pass) { raise_flag(P, kFlagWalkLimit); break; }
    // next trigger pop strictly after the last one
    KeyRef m = last; uint32_t m_trig = kNone;
    for (uint32_t i = beg; i < end; ++i)
      for (int ev = 0; ev < (BOTH_EVENTS ? 2 : 1); ++ev) {
        const Fire f = ev == 0 ? corner_fire(P, c, P.crn[i]) : corner_fire_second(P, c, P.crn[i]);
        if (f.trig == kNone || key_descends_from(P, f.trig, v)) continue;
        if (!first && !key_less(P, last, f.key)) continue;
        if (m_trig == kNone || key_less(P, f.key, m)) { m = f.key; m_trig = f.trig; }
      }
    if (m_trig == kNone) break;
    if (queued && !key_less(P, m, key_ref_of(e.key, e.keyd, v))) break;   // v pops before this trigger
    bool any = false;
    float ins_d = 0.0f;
    for (int pass = 0; pass < 2; ++pass)                  // faces of this pop, in the trigger's circulator order
      for (uint32_t i = beg; i < end; ++i) {
        const Corner k = P.crn[i];
        const Fire f = corner_fire(P, c, k);
        const bool fires = f.trig == m_trig || (BOTH_EVENTS && corner_fire_second(P, c, k).trig == m_trig);
        if (!fires || corner_first_for(k, m_trig) != (pass == 0)) continue;
        if (infl) {
          const InflCand u = infl_candidate(P.dist[k.v1], P.dist[k.v2], k.a, k.b, k.c, P.infl_max);
          if (e.d == 0.0f || !u.ok || !(u.u3tmp < e.d)) continue;       // :252, :271, :298
          e.d = u.u3tmp;                                                // :300
          e.pred = k.v1; e.cut = k.v2;                                  // supports of the last lowering update (vector field :302-309)
          if (u.requeue) { any = true; ins_d = e.d; }                   // :311 -> pq.insert(v, distances[v]) :451,459,467
        } else {
          const CvpUpd u = cvp_update(P.dist[k.v1], P.dist[k.v2], e.d, k.a, k.b, k.c);
          if (u.ok) {
            e.d = u.u3; e.pred = (u.sel == 1) ? k.v1 : k.v2; e.dir = u.dir; e.cut = corner_face(k);
            any = true; ins_d = e.d;
          }
        }
      }
    if (any && !mute) {                                    // ordinary pop, or a place inside the cascade of this trigger
      e.key = key_for(P, ins_d, v, m); e.keyd = ins_d; queued = true;
    }
    last = m; first = false;
  }
  if (!queued) { e.key = key_inf(); e.keyd = inf_f(); if (!infl) e.pred = v; }
  e.t = key_time(e.key);
  return e;
}
MNAV_HD Eval eval_cvp(const Plan& P, const Ctl& c, uint32_t v) { return eval_cvp_t<false>(P, c, v); }

// After convergence and verification: the vertices around the seed face once more, with both fire events of the faces that
// have a seed support (corner_fire_second).  Value and pop key are what they were -- the second event re-applies a candidate that
// was applied before --; predecessor, direction and cutting face are the last successful application's, like in the
// reference.  `v`: a support of one of the seeds' corners.  CVP planner only.
MNAV_HD void seed_ring_fix(const Plan& P, const Ctl& c, uint32_t v)
{
  if (v == kNone || v >= P.V || is_seed(P, v) || P.blocked[v] || !(P.dist[v] < inf_f())) return;
  const Eval e = eval_cvp_t<true>(P, c, v);
  if (f2u(e.d) != f2u(P.dist[v]) || e.key != P.tkey[v]) return;
  P.pred[v] = e.pred; P.dirn[v] = e.dir; P.cutf[v] = e.cut;
}

// ---------------------------------------------------------------------------------------
// eval_cvp in two parts, for the wide step kernel (k_step<cvp, true>, mnav.hip): what a vertex's incident faces
// contribute does not depend on the vertex's own state, so it is prepared face by face (one lane per face: fire event +
// float64 candidate, make_cvp_item) and the replay then runs per vertex over the prepared items (eval_cvp_items) -- the
// same decisions in the same order as eval_cvp, which stays the specification (tests/test_schedule_model.py holds the two
// against each other on every evaluation of the model).  CVP planner only (the inflation wave keeps eval_cvp).
// ---------------------------------------------------------------------------------------
struct CvpItem {
  unsigned long long hi;    // pop key of the trigger: PopKey {hi, up, lvl} ...
  uint32_t up, lvl;
  unsigned long long own;   // ... and its own (value, id) pair; the trigger's id is pair_id(own)
  double u3tmp, cand;       // CvpCand
  float dir;
  uint32_t meta;            // bit 0: the face can fire (and its trigger does not descend from the vertex), bits 1-2 sel, 3-4 kind,
};                          // bit 5: the face comes first in the trigger's circulator (corner_first_for)
static_assert(sizeof(CvpItem) == 48, "CvpItem: 48 bytes of LDS per incident face");
constexpr uint32_t kItemValid = 1u, kItemFirst = 32u;

// (t1, t2, d1, d2: pop keys and potentials of the face's two supports, loaded by the caller; CVP planner: keyd is not in use)
MNAV_HD CvpItem make_cvp_item_pre(const Plan& P, const Ctl& c, uint32_t v, const Corner& k, const PopKey& t1, const PopKey& t2, float d1, float d2)
{
  CvpItem it;
  it.hi = 0ull; it.up = kNone; it.lvl = 0u; it.own = 0ull; it.u3tmp = 0.0; it.cand = 0.0; it.dir = 0.0f; it.meta = 0u;
  if (k.v1 == kNone) return it;
  const Fire f = corner_fire_pre(P, c, k, key_ref_of(t1, d1, k.v1), key_ref_of(t2, d2, k.v2), d1, d2);
  if (f.trig == kNone || key_descends_from_pre(P, f.trig == k.v1 ? t1 : t2, v)) return it;          // (spec: eval_cvp)
  const CvpCand cd = cvp_candidate(d1, d2, k.a, k.b, k.c);
  it.hi = f.key.k.hi; it.up = f.key.k.up; it.lvl = f.key.k.lvl; it.own = f.key.own;
  it.u3tmp = cd.u3tmp; it.cand = cd.cand; it.dir = cd.dir;
  it.meta = kItemValid | ((uint32_t)cd.sel << 1) | ((uint32_t)cd.kind << 3) | (corner_first_for(k, f.trig) ? kItemFirst : 0u);
  return it;
}

MNAV_HD CvpItem make_cvp_item(const Plan& P, const Ctl& c, uint32_t v, const Corner& k)
{
  if (k.v1 == kNone) { const PopKey z = key_inf(); return make_cvp_item_pre(P, c, v, k, z, z, inf_f(), inf_f()); }
  return make_cvp_item_pre(P, c, v, k, P.tkey[k.v1], P.tkey[k.v2], P.dist[k.v1], P.dist[k.v2]);
}

// Items: field accessors hi(k), up(k), lvl(k), own(k), u3tmp(k), cand(k), dir(k), meta(k) of the vertex's k-th corner -- an array of
// CvpItem on the CPU (CvpItemArray), field-major arrays in LDS on the device (a lane then only reads the fields it looks at).
struct CvpItemArray {
  const CvpItem* p;
  MNAV_HD unsigned long long hi(uint32_t k) const { return p[k].hi; }
  MNAV_HD uint32_t up(uint32_t k) const { return p[k].up; }
  MNAV_HD uint32_t lvl(uint32_t k) const { return p[k].lvl; }
  MNAV_HD unsigned long long own(uint32_t k) const { return p[k].own; }
  MNAV_HD double u3tmp(uint32_t k) const { return p[k].u3tmp; }
  MNAV_HD double cand(uint32_t k) const { return p[k].cand; }
  MNAV_HD float dir(uint32_t k) const { return p[k].dir; }
  MNAV_HD uint32_t meta(uint32_t k) const { return p[k].meta; }
};

// The replay of eval_cvp over n prepared items.  `win` = index of the corner whose update set the returned value (kNone: none)
// and `win_sel` which of its supports is the predecessor; the caller resolves them to vertex / face ids from the corner record.
template <class Items>
MNAV_HD Eval eval_cvp_items(const Plan& P, uint32_t v, uint32_t n, const Items& item, uint32_t& win, int& win_sel)
{
  Eval e; e.d = inf_f(); e.t = inf_f(); e.key = key_inf(); e.pred = v; e.dir = 0.0f; e.cut = kNone; e.keyd = inf_f();
  win = kNone; win_sel = 0;
  KeyRef last = key_ref_of(key_inf(), inf_f(), 0);
  bool first = true, queued = false;
  const uint32_t max_pass = 2u * n + 2u;
  for (uint32_t pass_no = 0;; ++pass_no) {
    if (pass_no == max_pass) { raise_flag(P, kFlagWalkLimit); break; }
    KeyRef m = last; uint32_t m_trig = kNone;                         // next trigger pop strictly after the last one
    for (uint32_t k = 0; k < n; ++k) {
      if (!(item.meta(k) & kItemValid)) continue;
      const unsigned long long hi = item.hi(k);
      // decided by the main-front part of the key alone (the common case): no further field is read
      if (!first && hi < last.k.hi) continue;
      if (m_trig != kNone && hi > m.k.hi) continue;
      KeyRef fk; fk.k.hi = hi; fk.k.up = item.up(k); fk.k.lvl = item.lvl(k); fk.own = item.own(k);
      if (pair_id(fk.own) == m_trig) continue;                        // a second face of the current candidate's pop: the same key
      if (!first && !key_less(P, last, fk)) continue;
      if (m_trig == kNone || key_less(P, fk, m)) { m = fk; m_trig = pair_id(fk.own); }
    }
    if (m_trig == kNone) break;
    if (queued && !key_less(P, m, key_ref_of(e.key, e.keyd, v))) break;   // v pops before this trigger
    bool any = false;
    float ins_d = 0.0f;
    for (int pass = 0; pass < 2; ++pass)                              // faces of this pop, flagged face first
      for (uint32_t k = 0; k < n; ++k) {
        const uint32_t meta = item.meta(k);
        if (!(meta & kItemValid) || ((meta & kItemFirst) != 0u) != (pass == 0) || item.hi(k) != m.k.hi || pair_id(item.own(k)) != m_trig) continue;
        CvpCand cd; cd.u3tmp = item.u3tmp(k); cd.cand = item.cand(k); cd.dir = item.dir(k); cd.sel = (int)((meta >> 1) & 3u); cd.kind = (int)((meta >> 3) & 3u);
        int sel = 0; float dir = 0.0f;
        if (cvp_apply(cd, e.d, sel, dir)) { win = k; win_sel = sel; e.dir = dir; any = true; ins_d = e.d; }
      }
    if (any) { e.key = key_for(P, ins_d, v, m); e.keyd = ins_d; queued = true; }
    last = m; first = false;
  }
  if (!queued) { e.key = key_inf(); e.keyd = inf_f(); win = kNone; }
  e.t = key_time(e.key);
  return e;
}

// The same replay for vertices with at most kFastItems faces (nearly all of them), with the fields the loops look at held in
// registers: every loop is a fully unrolled, predicated sweep over eight slots, and LDS is only read where an update is applied.
// Ties in the main-front part of the key between DIFFERENT triggers (cascades) need key_less's walk: the replay then starts over
// on the general path.  Returns false in that case.
constexpr uint32_t kFastItems = 8;
template <class Items>
MNAV_HD bool eval_cvp_items_fast(const Plan& P, uint32_t v, uint32_t n, const Items& item, Eval& e, uint32_t& win, int& win_sel)
{
  unsigned long long hi[kFastItems]; uint32_t trig[kFastItems], meta[kFastItems];
MNAV_UNROLL
  for (uint32_t k = 0; k < kFastItems; ++k) {
    meta[k] = 0u; hi[k] = 0ull; trig[k] = kNone;
    if (k < n) { meta[k] = item.meta(k); hi[k] = item.hi(k); trig[k] = pair_id(item.own(k)); }
    if (!(meta[k] & kItemValid)) { meta[k] = 0u; trig[k] = kNone; }
  }
  e.d = inf_f(); e.t = inf_f(); e.key = key_inf(); e.pred = v; e.dir = 0.0f; e.cut = kNone; e.keyd = inf_f();
  win = kNone; win_sel = 0;
  unsigned long long last_hi = 0ull; uint32_t last_trig = kNone;
  bool first = true, queued = false;
  for (uint32_t pass_no = 0; pass_no <= kFastItems; ++pass_no) {       // (every pass consumes one trigger: at most n passes)
    unsigned long long m_hi = ~0ull; uint32_t m_trig = kNone; uint32_t m_k = 0;
    bool tie = false;
MNAV_UNROLL
    for (uint32_t k = 0; k < kFastItems; ++k) {
      if (trig[k] == kNone || trig[k] == last_trig) continue;
      if (!first && hi[k] < last_hi) continue;
      if (!first && hi[k] == last_hi) { tie = true; continue; }        // another trigger inside the last one's cascade
      if (m_trig != kNone && trig[k] == m_trig) continue;
      if (m_trig != kNone && hi[k] == m_hi) { tie = true; continue; }
      if (hi[k] < m_hi) { m_hi = hi[k]; m_trig = trig[k]; m_k = k; }
    }
    if (tie) return false;
    if (m_trig == kNone) break;
    KeyRef m; m.k.hi = m_hi; m.k.up = item.up(m_k); m.k.lvl = item.lvl(m_k); m.own = item.own(m_k);
    if (queued && !key_less(P, m, key_ref_of(e.key, e.keyd, v))) break;   // v pops before this trigger
    bool any = false;
    float ins_d = 0.0f;
MNAV_UNROLL
    for (int pass = 0; pass < 2; ++pass)
MNAV_UNROLL
      for (uint32_t k = 0; k < kFastItems; ++k) {
        if (trig[k] != m_trig || hi[k] != m_hi || ((meta[k] & kItemFirst) != 0u) != (pass == 0)) continue;
        CvpCand cd; cd.u3tmp = item.u3tmp(k); cd.cand = item.cand(k); cd.dir = item.dir(k); cd.sel = (int)((meta[k] >> 1) & 3u); cd.kind = (int)((meta[k] >> 3) & 3u);
        int sel = 0; float dir = 0.0f;
        if (cvp_apply(cd, e.d, sel, dir)) { win = k; win_sel = sel; e.dir = dir; any = true; ins_d = e.d; }
      }
    if (any) { e.key = key_for(P, ins_d, v, m); e.keyd = ins_d; queued = true; }
    last_hi = m_hi; last_trig = m_trig; first = false;
  }
  if (!queued) { e.key = key_inf(); e.keyd = inf_f(); win = kNone; }
  e.t = key_time(e.key);
  return true;
}

template <class Items>
MNAV_HD Eval eval_cvp_items_any(const Plan& P, uint32_t v, uint32_t n, const Items& item, uint32_t& win, int& win_sel)
{
  Eval e;
  if (n <= kFastItems && eval_cvp_items_fast(P, v, n, item, e, win, win_sel)) return e;
  return eval_cvp_items(P, v, n, item, win, win_sel);
}

// the two parts put together on one thread (CPU model; the kernel's fall-back for vertices of very high valence is eval_cvp)
constexpr uint32_t kCvpItemsMax = 64;
MNAV_HD Eval eval_cvp_two_part(const Plan& P, const Ctl& c, uint32_t v)
{
  const uint32_t beg = P.crn_ptr[v], n = P.crn_ptr[v + 1] - beg;
  if (n > kCvpItemsMax || P.seed_mask) return eval_cvp(P, c, v);
  CvpItem items[kCvpItemsMax];
  for (uint32_t k = 0; k < n; ++k) items[k] = make_cvp_item(P, c, v, P.crn[beg + k]);
  uint32_t win; int sel;
  CvpItemArray arr; arr.p = items;
  Eval e = eval_cvp_items_any(P, v, n, arr, win, sel);
  if (win != kNone) { const Corner k = P.crn[beg + win]; e.pred = (sel == 1) ? k.v1 : k.v2; e.cut = corner_face(k); }
  return e;
}

// ---------------------------------------------------------------------------------------
// One work-list entry.  `Ops` supplies: push(v) (dedup'd append to the next list), push_dirty(u)
// (the same, and marks u as "a neighbour moved" for the next step),
// note_changed(), note_cut(float) (pop time of an in-band vertex that moved), note_min(float), note_eval(), park(v, t) (dedup'd append to the waiting list + note_min(t)).
// ---------------------------------------------------------------------------------------
// R = state the rule reads, W = state it writes.  The kernels use R == W (in-place, racy but
// monotone towards the fixed point); the CPU model can also run it Jacobi-style on a snapshot to
// emulate the worst interleaving of concurrently evaluated neighbours.
template <uint32_t PLANNER, class Ops>
MNAV_HD void process_entry_rw(const Plan& P, const Plan& W, const Ctl& c, uint32_t v, Ops& ops)
{
  if (is_seed(P, v)) return;                                     // seeds are fixed from the start
  constexpr bool cvp = (PLANNER == kPlannerCvp);
  PopKey old_key = key_inf();
  if constexpr (cvp) old_key = P.tkey[v];
  const float old_t = cvp ? key_time(old_key) : P.dist[v];
  if (old_t < c.thr_fixed) return;                               // settled by an earlier band
  if (cvp && P.blocked[v]) return;                               // never updated (cvp :802,825,848)
  // parked out of band, and no neighbour moved since it was evaluated: keep waiting, as is
  if (!c.band_new && !(old_t < c.thr) && old_t < inf_f() && P.dirty[v] != (uint32_t)c.it) {
    ops.park(v, old_t);
    return;
  }
  ops.note_eval();
  Eval e;
  if constexpr (cvp) e = eval_cvp(P, c, v); else e = eval_dijkstra(P, c, v);
#ifdef MNAV_CHECK_TWO_PART                // CPU model: the wide step kernel's two-part evaluation against the specification, on every evaluation
  if constexpr (cvp) {
    const Eval w = eval_cvp_two_part(P, c, v);
    if (f2u(w.d) != f2u(e.d) || f2u(w.t) != f2u(e.t) || w.key != e.key || w.pred != e.pred || f2u(w.dir) != f2u(e.dir) || w.cut != e.cut ||
        f2u(w.keyd) != f2u(e.keyd)) {
      fprintf(stderr, "two-part CVP evaluation differs from eval_cvp at vertex %u: d %.9g / %.9g pred %u / %u cut %u / %u\n", v, w.d, e.d, w.pred, e.pred, w.cut, e.cut);
      abort();
    }
  }
#endif
  const float old_d = P.dist[v];
  bool changed = (f2u(e.d) != f2u(old_d)) || (f2u(e.t) != f2u(old_t));
  if (cvp) changed = changed || (e.key != old_key) || (e.pred != P.pred[v]) || (e.cut != P.cutf[v]) || (f2u(e.dir) != f2u(P.dirn[v])) ||
                     (P.keyd && f2u(e.keyd) != f2u(P.keyd[v]));
  else changed = changed || (e.pred != P.pred[v]);
  if (changed) {
    W.dist[v] = e.d; W.pred[v] = e.pred;
    if constexpr (cvp) { W.tkey[v] = e.key; W.dirn[v] = e.dir; W.cutf[v] = e.cut; if (W.keyd) W.keyd[v] = e.keyd; }
  }
  if constexpr (cvp) {
    // the rule reads v's own stored key (key_descends_from): a cascade member whose key moved looks again
    if ((e.key.lvl > 0u || old_key.lvl > 0u) && e.key != old_key) { ops.push_dirty(v); ops.note_changed(); }
  }
  const bool was_in = old_t < c.thr, now_in = e.t < c.thr;
  if ((changed && (was_in || now_in)) || (now_in && c.band_new)) {
    // v is (or was) usable by its neighbours and moved, or only just entered the band:
    // the neighbours must look again, and the band cannot complete in this step.
    ops.note_changed();
    if (was_in != now_in) ops.note_cut(fminf(old_t, e.t));      // moved across the band's upper bound (kCutAfter)
    if (cvp) {
      for (uint32_t i = P.crn_ptr[v]; i < P.crn_ptr[v + 1]; ++i) {
        const Corner k = P.crn[i];
        if (k.v1 == kNone) continue;
        ops.push_dirty(k.v1); ops.push_dirty(k.v2);
      }
    } else {
      for (uint32_t i = P.row_ptr[v]; i < P.row_ptr[v + 1]; ++i) ops.push_dirty(P.nbr[i].u);
    }
  }
  if (!now_in && e.t < inf_f()) {
    ops.park(v, e.t);                                            // keyed, waits for its band
  }
}

template <uint32_t PLANNER, class Ops>
MNAV_HD void process_entry(const Plan& P, const Ctl& c, uint32_t v, Ops& ops) { process_entry_rw<PLANNER>(P, P, c, v, ops); }

// Repair sweep entry (one per vertex, step with ctl.repair == 1): see controller().
template <uint32_t PLANNER, class Ops>
MNAV_HD void process_repair(const Plan& P, const Ctl& c, uint32_t v, Ops& ops)
{
  if (is_seed(P, v)) return;
  constexpr bool cvp = (PLANNER == kPlannerCvp);
  float d = P.dist[v];
  if (!(d < inf_f())) return;
  float t = d;
  if constexpr (cvp) t = key_time(P.tkey[v]);
  // CVP: a vertex whose value was set by a trigger beyond goal_dist pops after that trigger, i.e. its
  // POP TIME (not necessarily its value: non-causal updates undercut the front) lies above goal_dist
  if (t > c.goal_dist) {
    ops.note_eval();
    Eval e;
    if constexpr (cvp) e = eval_cvp(P, c, v); else e = eval_dijkstra(P, c, v);
    if constexpr (cvp) {
      // such vertices can support each other (in pop order): sweep again until nothing moves
      if (f2u(e.d) != f2u(d) || e.key != P.tkey[v]) ops.note_changed();
    }
    P.dist[v] = e.d; P.pred[v] = e.pred;
    if constexpr (cvp) { P.tkey[v] = e.key; P.dirn[v] = e.dir; P.cutf[v] = e.cut; if (P.keyd) P.keyd[v] = e.keyd; }
    d = e.d; t = e.t;
  }
  if (t >= c.thr && t < inf_f()) ops.park(v, t);
}

// Band cut (step with ctl.repair == 3, c.thr = the cut): nothing is evaluated.  Every keyed vertex at or above the cut
// goes to the (fresh) waiting list -- among them those that were in band until now -- so that the first step of the
// restarted band looks at all of them; the caller carries the work list over (push_dirty of every entry).
template <uint32_t PLANNER, class Ops>
MNAV_HD void process_cut(const Plan& P, const Ctl& c, uint32_t v, Ops& ops)
{
  if (is_seed(P, v)) return;
  float t = P.dist[v];
  if constexpr (PLANNER == kPlannerCvp) t = key_time(P.tkey[v]);
  if (t >= c.thr && t < inf_f()) ops.park(v, t);
}

// Work-list rebuild after a band shrink (step with ctl.repair == 2): every keyed vertex that is
// not settled yet is re-evaluated under the narrower band and re-enters the list.
// ---------------------------------------------------------------------------------------
// The exact band (Ctl.exact_wanted): the reference's own procedure for ONE band -- pop the vertices of the band one at a time, in
// key order; a vertex acts as a support only once it has popped, i.e. once its state is final.  No provisional state ever feeds
// another, so there is nothing to iterate and nothing that can cycle: every pop fixes one vertex for good.  The routine
//   1. resets the band: every vertex with a pop time in [thr_fixed, thr) and its corner neighbours that are not settled are
//      evaluated again from SETTLED supports only (exact_entry with bound_v = kNone);
//   2. repeats: the candidate with the smallest key pops (bound_v = that vertex), its corner neighbours are evaluated again;
//      until no candidate with a pop time below thr is left;
//   3. hands back to the band steps: force_cut (a cut step at thr rebuilds the waiting list from the vertices' stored keys), then
//      the band starts again on a state that is a fixed point of the step rule -- it completes at once.
// Candidates are kept in a plain list (a plan's idle work-list buffer), the minimum is found by scanning it: a band that needs
// this is a few hundred vertices.  Host (CPU model) and device (k_exact_band: one workgroup, lanes over the list / the neighbours)
// share the pieces below.  Inflation waves only (the CVP planner's bands have never needed it).
// ---------------------------------------------------------------------------------------
constexpr uint32_t kExactStamp = 0xFFFFFFF1u;                        // Plan.stamp value of a vertex on the routine's candidate list
MNAV_HD Ctl exact_ctl(const Ctl& c) { Ctl x = c; x.exact = 1u; x.bound_v = kNone; return x; }
// evaluates v on final supports only and stores its state; true: v is a candidate now (keyed, not popped, pop time below thr)
MNAV_HD bool exact_entry(const Plan& P, const Ctl& x, uint32_t v)
{
  if (v == kNone || v >= P.V || is_seed(P, v) || P.blocked[v]) return false;
  if (popped_exact(P, x, key_time(P.tkey[v]), key_ref(P, v))) return false;   // final
  const Eval e = eval_cvp(P, x, v);
  P.dist[v] = e.d; P.pred[v] = e.pred; P.tkey[v] = e.key; P.dirn[v] = e.dir; P.cutf[v] = e.cut;
  if (P.keyd) P.keyd[v] = e.keyd;
  return e.t < x.thr;
}
// the same in two halves, for lanes that work side by side: all evaluate (reads only), barrier, all store
struct ExactEval { Eval e; bool act; };
MNAV_HD ExactEval exact_eval(const Plan& P, const Ctl& x, uint32_t v)
{
  ExactEval r; r.act = false;
  if (v == kNone || v >= P.V || is_seed(P, v) || P.blocked[v]) return r;
  if (popped_exact(P, x, key_time(P.tkey[v]), key_ref(P, v))) return r;
  r.e = eval_cvp(P, x, v); r.act = true;
  return r;
}
MNAV_HD bool exact_store(const Plan& P, const Ctl& x, uint32_t v, const ExactEval& r)
{
  if (!r.act) return false;
  P.dist[v] = r.e.d; P.pred[v] = r.e.pred; P.tkey[v] = r.e.key; P.dirn[v] = r.e.dir; P.cutf[v] = r.e.cut;
  if (P.keyd) P.keyd[v] = r.e.keyd;
  return r.e.t < x.thr;
}
// is candidate u still one (its state may have moved since it was listed), and does it pop before the best so far?
MNAV_HD bool exact_better(const Plan& P, const Ctl& x, uint32_t u, uint32_t best)
{
  const KeyRef ku = key_ref(P, u);
  const float t = key_time(ku.k);
  if (!(t < x.thr) || popped_exact(P, x, t, ku)) return false;
  return best == kNone || key_less(P, ku, key_ref(P, best));
}
// the state the band steps resume from
MNAV_HD Ctl exact_done_ctl(const Ctl& c) { Ctl q = c; q.exact = 0u; q.bound_v = kNone; q.exact_wanted = 0u; q.force_cut = 1u; return q; }

template <uint32_t PLANNER, class Ops>
MNAV_HD void process_rebuild(const Plan& P, const Ctl& c, uint32_t v, Ops& ops)
{
  if (!(P.dist[v] < inf_f())) return;
  process_entry<PLANNER>(P, c, v, ops);      // c.band_new == 1: in-band vertices wake their neighbours
}

// ---------------------------------------------------------------------------------------
// The inflation layer's repulsive vector field (vector_map_, inflation_layer.cpp:277-309), from the converged wave.
// Two parts.  (A) Faces with two lethal corners and one free one add their direction to the vectors of all three
// vertices, `vec = normalized(vec + dir)`, every time the face is visited: at the pops of both lethal corners (all at
// time 0, in (0, id) order, before any free vertex pops), twice per pop (:423-427) -- an order-dependent float
// accumulation, but each vertex's sequence only involves its own faces: infl_accumulate replays it.  (B) A later
// update that lowers a free vertex from supports that are not both lethal ASSIGNS its vector from the supports' (:302
// -309); the last lowering update wins, and eval_cvp records its supports in pred / cutf: infl_assign.
// ---------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
MNAV_HD V3 v3_of(const float* p) { V3 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; return r; }
MNAV_HD V3 v3_add(V3 a, V3 b) { V3 r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; return r; }
MNAV_HD V3 v3_sub(V3 a, V3 b) { V3 r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; return r; }
MNAV_HD V3 v3_mul(V3 a, float s) { V3 r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; return r; }
MNAV_HD V3 v3_unit(V3 a)                                           // lvr2 BaseVector::normalized(): divide by the length
{
  const float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
  V3 r; r.x = a.x / l; r.y = a.y / l; r.z = a.z / l;
  return r;
}
MNAV_HD bool infl_is_lethal(uint8_t m) { return m == kInflSeed || m == kInflSeedMute; }

constexpr int kInflMaxEvents = 64;
// (A) for vertex h.  crn / crn_walk: the inflation corner table (edge distances) and the walk positions of
// HostTopology::crn_walk.  Returns 0 = no such face around h, 1 = out holds the vector, -1 = cannot be replayed here
// (a vertex with too many neighbours for the 5-bit positions, or more than kInflMaxEvents visits).
MNAV_HD int infl_accumulate(const Plan& P, const uint32_t* crn_walk, const float* xyz, uint32_t h, float out[3])
{
  unsigned long long key[kInflMaxEvents];
  V3 dir[kInflMaxEvents];
  int n = 0;
  const uint8_t mh = P.seed_mask[h];
  for (uint32_t i = P.crn_ptr[h]; i < P.crn_ptr[h + 1]; ++i) {
    const Corner k = P.crn[i];
    if (k.v1 == kNone) continue;
    const uint8_t m1 = P.seed_mask[k.v1], m2 = P.seed_mask[k.v2];
    const int nl = (infl_is_lethal(m1) ? 1 : 0) + (infl_is_lethal(m2) ? 1 : 0) + (infl_is_lethal(mh) ? 1 : 0);
    if (nl != 2) continue;
    // the free corner and, in the cyclic order (v1, v2, h), its two predecessors = the update's (v1, v2) (:444-470)
    uint32_t fv, s1, s2; float a, b, c;                              // a = |s2 fv|, b = |s1 fv|, c = |s1 s2|
    if (!infl_is_lethal(mh)) { fv = h; s1 = k.v1; s2 = k.v2; a = k.a; b = k.b; c = k.c; }
    else if (!infl_is_lethal(m1)) { fv = k.v1; s1 = k.v2; s2 = h; a = k.b; b = k.c; c = k.a; }
    else { fv = k.v2; s1 = h; s2 = k.v1; a = k.c; b = k.a; c = k.b; }
    const InflCand u = infl_candidate(0.0f, 0.0f, a, b, c, P.infl_max);
    if (!u.ok) continue;                                             // :271 comes before the vector part
    const V3 p3 = v3_of(xyz + 3 * (size_t)fv), p1 = v3_of(xyz + 3 * (size_t)s1), p2 = v3_of(xyz + 3 * (size_t)s2);
    const V3 d = v3_unit(v3_add(v3_sub(p3, p2), v3_sub(p3, p1)));    // :282
    const uint32_t walk = crn_walk[i];
    if (walk == 0xFFFFFFFFu) return -1;
    const uint32_t xs[3] = { k.v1, k.v2, h };
    const uint8_t ms[3] = { m1, m2, mh };
    for (int q = 0; q < 3; ++q) {
      if (ms[q] != kInflSeed) continue;                              // a free corner does not pop here; an invalid lethal one is skipped (:417)
      const uint32_t pa = (walk >> (10 * q)) & 31u, pb = (walk >> (10 * q + 5)) & 31u;
      for (int t = 0; t < (pa == pb ? 1 : 2); ++t) {
        if (n == kInflMaxEvents) return -1;
        key[n] = ((unsigned long long)xs[q] << 5) | (t == 0 ? pa : pb);
        dir[n] = d;
        ++n;
      }
    }
  }
  if (n == 0) return 0;
  for (int i = 1; i < n; ++i) {                                      // visit order: pop (vertex id), then position in its walk
    const unsigned long long kk = key[i]; const V3 dd = dir[i];
    int j = i - 1;
    for (; j >= 0 && key[j] > kk; --j) { key[j + 1] = key[j]; dir[j + 1] = dir[j]; }
    key[j + 1] = kk; dir[j + 1] = dd;
  }
  V3 v; v.x = 0.0f; v.y = 0.0f; v.z = 0.0f;
  for (int i = 0; i < n; ++i) v = v3_unit(v3_add(v, dir[i]));        // :293-295
  out[0] = v.x; out[1] = v.y; out[2] = v.z;
  return 1;
}

// (B) for a free vertex w with a distance: 0 = keeps what (A) gave it (its last lowering update had two lethal supports,
// or it was never lowered), 1 = out holds the assigned vector, 2 = a support's vector is not final yet (sweep again).
// state[u]: 0 unknown, 1 final with vector, 2 final without (value_or(zero) :306-307).
MNAV_HD int infl_assign(const Plan& P, const float* vec, const uint8_t* state, uint32_t w, float out[3])
{
  const float d = P.dist[w];
  const uint32_t s1 = P.pred[w], s2 = P.cutf[w];
  if (!(d < inf_f()) || s1 == w || s2 == kNone) return 0;
  const float u1 = P.dist[s1], u2 = P.dist[s2];
  if (u1 == 0.0f && u2 == 0.0f) return 0;                            // :302
  if (state[s1] == 0 || state[s2] == 0) return 2;
  V3 va, vb; va.x = va.y = va.z = 0.0f; vb = va;
  if (state[s1] == 1) va = v3_of(vec + 3 * (size_t)s1);
  if (state[s2] == 1) vb = v3_of(vec + 3 * (size_t)s2);
  const float d31 = d - u1, d32 = d - u2;                            // :274-275
  const V3 r = v3_unit(v3_add(v3_mul(va, d31), v3_mul(vb, d32)));    // :308
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
  return 1;
}

// Verification sweep entry (k_cvp_verify): on the CONVERGED state every vertex must be a fixed point of the replay
// rule.  The work-list iteration notifies the corner neighbours of a vertex that moved; a vertex whose pop key hangs
// off a FAR cascade ancestor (key_for climbs the tree) is not told when that ancestor's place in the tree changes, so
// a few deep-cascade members can be left with a stale key.  The sweep finds them; with `fix` it stores the
// re-evaluated state, and the caller sweeps again until a sweep finds nothing (only then is the result returned).
MNAV_HD bool verify_entry(const Plan& P, const Ctl& c, uint32_t v, const Eval& e, bool fix)
{
  const bool same = f2u(e.d) == f2u(P.dist[v]) && e.key == P.tkey[v] && e.pred == P.pred[v] &&
                    (!(e.d < inf_f()) || (e.cut == P.cutf[v] && f2u(e.dir) == f2u(P.dirn[v]))) &&
                    (!P.keyd || f2u(e.keyd) == f2u(P.keyd[v]));
  if (!same && fix) {
    P.dist[v] = e.d; P.pred[v] = e.pred; P.tkey[v] = e.key; P.dirn[v] = e.dir; P.cutf[v] = e.cut;
    if (P.keyd) P.keyd[v] = e.keyd;
  }
  (void)c;
  return same;
}
constexpr int kVerifySweeps = 8;         // fixing sweeps before the plan is given up as INTERNAL_ERROR

}  // namespace mnav
