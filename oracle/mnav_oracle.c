/*
 * mnav_oracle.c -- CPU restatement of the mesh_navigation wavefront planners.
 *
 * TEST INFRASTRUCTURE ONLY (see mnav_oracle.h).  PARITY UNPINNED: the reference
 * holds no planner tests/golden vectors and cannot be built here; the only
 * reference known-answer (InflationLayer single triangle) is replayed in
 * tests/test_oracle_kat.py.
 *
 * Single-threaded, compiled like the reference's Release build (-O2/-O3, no
 * -march, no -ffast-math; we add -ffp-contract=off so that no FMA can ever be
 * formed).  Every function cites the reference file:line it follows; paths are
 * relative to the reference checkout.
 */
#include "mnav_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NONE 0xFFFFFFFFu

/* ------------------------------------------------------------------------- */
/* float vector helpers: lvr2::BaseVector<float> semantics (CONVENTION: lvr2  */
/* is un-vendored; plain float component arithmetic, length = sqrtf(dot),     */
/* normalize = component / length).                                           */
/* ------------------------------------------------------------------------- */
typedef struct { float x, y, z; } vec3;

static vec3 v3(float x, float y, float z) { vec3 r = { x, y, z }; return r; }
static vec3 v3_load(const float* p) { return v3(p[0], p[1], p[2]); }
static void v3_store(float* p, vec3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static vec3 v3_add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static vec3 v3_sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static vec3 v3_scale(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static vec3 v3_div(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static float v3_dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static vec3 v3_cross(vec3 a, vec3 b)
{
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static float v3_length2(vec3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static float v3_length(vec3 a) { return sqrtf(v3_length2(a)); }
static vec3 v3_normalized(vec3 a) { return v3_div(a, v3_length(a)); }
static float v3_distance2(vec3 a, vec3 b) { return v3_length2(v3_sub(a, b)); }
static float v3_distance(vec3 a, vec3 b) { return sqrtf(v3_distance2(a, b)); }

/* lvr2::BaseVector::rotated(axis, angle) stand-in (used at
 * cvp_mesh_planner.cpp:234).  CONVENTION: Rodrigues rotation about the unit
 * axis n by angle radians, float arithmetic with cosf/sinf:
 *   v' = v cos + (n x v) sin + n (n.v)(1 - cos)                              */
static vec3 v3_rotated(vec3 v, vec3 n, float angle)
{
  const float c = cosf(angle);
  const float s = sinf(angle);
  const vec3 nxv = v3_cross(n, v);
  const float ndv = v3_dot(n, v);
  const float k = ndv * (1.0f - c);
  return v3(v.x * c + nxv.x * s + n.x * k, v.y * c + nxv.y * s + n.y * k,
            v.z * c + nxv.z * s + n.z * k);
}

static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ------------------------------------------------------------------------- */
/* mesh topology                                                              */
/* ------------------------------------------------------------------------- */
struct mo_mesh {
  uint32_t V, F, E;
  int manifold;        /* 0: pmp would reject faces of this list; rows stay in ascending id */
  float* xyz;          /* V*3 */
  uint32_t* fv;        /* F*3 */
  uint32_t* fe;        /* F*3: edge between fv[k] and fv[(k+1)%3] */
  uint32_t* ev;        /* E*2 */
  uint32_t* ef;        /* E*2 incident faces (NONE if boundary), in id order */
  uint32_t* ve_ptr;    /* V+1 */
  uint32_t* ve;        /* 2E edge ids around vertex, in half-edge circulator order (see he_build) */
  uint32_t* vf_ptr;    /* V+1 */
  uint32_t* vf;        /* 3F face ids around vertex, in half-edge circulator order */
};

static uint64_t edge_key(uint32_t a, uint32_t b)
{
  return a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
}

static uint64_t hash64(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

/* ------------------------------------------------------------------------- */
/* Circulator order.  lvr2::PMPMesh wraps a pmp::SurfaceMesh; getEdgesOfVertex /
 * getFacesOfVertex walk the outgoing half-edges of a vertex counter-clockwise,
 * starting at the vertex's stored outgoing half-edge.  Which half-edge that is
 * depends on the order in which the faces were added (pmp::SurfaceMesh::
 * add_face, the OpenMesh algorithm, incl. adjust_outgoing_halfedge), so the
 * mesh is rebuilt here face by face exactly like PMPMesh(MeshBufferPtr) does
 * (mesh_map.cpp:273).  The order matters: CVP applies the faces of a popped
 * vertex in this order and its update is not a pure minimum on cost-inflated
 * triangles; vertex normals are summed in it; searchNeighbourFaces walks it.
 * lvr2/pmp are un-vendored: this restates the published pmp algorithm and is
 * checked against oracle/_ref (the reference's own code on the same model).
 * Half-edge 2e goes ev[2e] -> ev[2e+1] (new_edge(start,end)), 2e+1 back.      */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint32_t* vh;      /* V: outgoing half-edge of a vertex */
  uint32_t* hto;     /* H: to-vertex */
  uint32_t* hface;   /* H: face left of the half-edge, NONE on the boundary */
  uint32_t* hnext;   /* H */
  uint32_t* hprev;   /* H */
  uint32_t* fh;      /* F: half-edge of a face */
  uint32_t H, Hcap;
} he_mesh;

static uint32_t he_opp(uint32_t h) { return h ^ 1u; }
static uint32_t he_cw(const he_mesh* m, uint32_t h) { return m->hnext[he_opp(h)]; }
static uint32_t he_ccw(const he_mesh* m, uint32_t h) { return he_opp(m->hprev[h]); }
static int he_vertex_is_boundary(const he_mesh* m, uint32_t v)
{
  const uint32_t h = m->vh[v];
  return !(h != NONE && m->hface[h] != NONE);
}
static uint32_t he_find(const he_mesh* m, uint32_t start, uint32_t end)
{
  uint32_t h = m->vh[start];
  const uint32_t hh = h;
  if (h != NONE) {
    do {
      if (m->hto[h] == end) return h;
      h = he_cw(m, h);
    } while (h != hh);
  }
  return NONE;
}
static void he_set_next(he_mesh* m, uint32_t h, uint32_t nh) { m->hnext[h] = nh; m->hprev[nh] = h; }
static void he_adjust_outgoing(he_mesh* m, uint32_t v)
{
  uint32_t h = m->vh[v];
  const uint32_t hh = h;
  if (h != NONE) {
    do {
      if (m->hface[h] == NONE) { m->vh[v] = h; return; }
      h = he_cw(m, h);
    } while (h != hh);
  }
}
/* returns 0 on a topological error (complex vertex / complex edge / patch re-linking failed) */
static int he_add_triangle(he_mesh* m, uint32_t f, const uint32_t vs[3])
{
  uint32_t hs[3]; int is_new[3], needs_adjust[3] = { 0, 0, 0 };
  uint32_t nc[18][2]; int n_nc = 0;
  for (int i = 0; i < 3; ++i) {
    const int ii = (i + 1) % 3;
    if (!he_vertex_is_boundary(m, vs[i])) return 0;
    hs[i] = he_find(m, vs[i], vs[ii]);
    is_new[i] = hs[i] == NONE;
    if (!is_new[i] && m->hface[hs[i]] != NONE) return 0;
  }
  for (int i = 0; i < 3; ++i) {
    const int ii = (i + 1) % 3;
    if (!is_new[i] && !is_new[ii]) {
      const uint32_t inner_prev = hs[i], inner_next = hs[ii];
      if (m->hnext[inner_prev] != inner_next) {
        const uint32_t outer_prev = he_opp(inner_next);
        uint32_t boundary_prev = outer_prev;
        do {
          boundary_prev = he_opp(m->hnext[boundary_prev]);
        } while (m->hface[boundary_prev] != NONE || boundary_prev == inner_prev);
        const uint32_t boundary_next = m->hnext[boundary_prev];
        if (boundary_next == inner_next) return 0;
        const uint32_t patch_start = m->hnext[inner_prev], patch_end = m->hprev[inner_next];
        nc[n_nc][0] = boundary_prev; nc[n_nc++][1] = patch_start;
        nc[n_nc][0] = patch_end; nc[n_nc++][1] = boundary_next;
        nc[n_nc][0] = inner_prev; nc[n_nc++][1] = inner_next;
      }
    }
  }
  for (int i = 0; i < 3; ++i) {
    const int ii = (i + 1) % 3;
    if (is_new[i]) {                                /* new_edge(vs[i], vs[ii]) */
      const uint32_t h0 = m->H, h1 = m->H + 1;
      m->H += 2;
      m->hto[h0] = vs[ii]; m->hto[h1] = vs[i];
      m->hface[h0] = m->hface[h1] = NONE; m->hnext[h0] = m->hnext[h1] = NONE; m->hprev[h0] = m->hprev[h1] = NONE;
      hs[i] = h0;
    }
  }
  m->fh[f] = hs[2];
  for (int i = 0; i < 3; ++i) {
    const int ii = (i + 1) % 3;
    const uint32_t v = vs[ii];
    const uint32_t inner_prev = hs[i], inner_next = hs[ii];
    const int id = (is_new[i] ? 1 : 0) | (is_new[ii] ? 2 : 0);
    if (id) {
      const uint32_t outer_prev = he_opp(inner_next), outer_next = he_opp(inner_prev);
      if (id == 1) {
        const uint32_t boundary_prev = m->hprev[inner_next];
        nc[n_nc][0] = boundary_prev; nc[n_nc++][1] = outer_next;
        m->vh[v] = outer_next;
      } else if (id == 2) {
        const uint32_t boundary_next = m->hnext[inner_prev];
        nc[n_nc][0] = outer_prev; nc[n_nc++][1] = boundary_next;
        m->vh[v] = boundary_next;
      } else {
        if (m->vh[v] == NONE) {
          m->vh[v] = outer_next;
          nc[n_nc][0] = outer_prev; nc[n_nc++][1] = outer_next;
        } else {
          const uint32_t boundary_next = m->vh[v];
          const uint32_t boundary_prev = m->hprev[boundary_next];
          nc[n_nc][0] = boundary_prev; nc[n_nc++][1] = outer_next;
          nc[n_nc][0] = outer_prev; nc[n_nc++][1] = boundary_next;
        }
      }
      nc[n_nc][0] = inner_prev; nc[n_nc++][1] = inner_next;
    } else {
      needs_adjust[ii] = (m->vh[v] == inner_next);
    }
    m->hface[hs[i]] = f;
  }
  for (int k = 0; k < n_nc; ++k) he_set_next(m, nc[k][0], nc[k][1]);
  for (int i = 0; i < 3; ++i) if (needs_adjust[i]) he_adjust_outgoing(m, vs[i]);
  return 1;
}

/* Re-orders the rows of ve / vf into circulator order.  Returns 0 if the face list is not a manifold
 * in pmp's sense (the reference's loader would discard faces and re-index, mesh_map.cpp:276-300). */
static int he_circulator_orders(mo_mesh* m)
{
  he_mesh h;
  const size_t Hcap = 2 * (size_t)m->E + 2;
  h.H = 0; h.Hcap = (uint32_t)Hcap;
  h.vh = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)m->V + 1));
  memset(h.vh, 0xFF, sizeof(uint32_t) * ((size_t)m->V + 1));
  h.hto = (uint32_t*)malloc(sizeof(uint32_t) * Hcap); h.hface = (uint32_t*)malloc(sizeof(uint32_t) * Hcap);
  h.hnext = (uint32_t*)malloc(sizeof(uint32_t) * Hcap); h.hprev = (uint32_t*)malloc(sizeof(uint32_t) * Hcap);
  h.fh = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)m->F + 1));
  int ok = 1;
  for (uint32_t f = 0; f < m->F && ok; ++f) ok = he_add_triangle(&h, f, m->fv + 3 * (size_t)f);
  if (ok && h.H != 2 * m->E) ok = 0;
  if (ok) {
    for (uint32_t v = 0; v < m->V; ++v) {
      uint32_t ne = 0, nf = 0;
      uint32_t x = h.vh[v];
      const uint32_t xx = x;
      if (x == NONE) continue;
      do {
        m->ve[m->ve_ptr[v] + ne++] = x >> 1;
        if (h.hface[x] != NONE) m->vf[m->vf_ptr[v] + nf++] = h.hface[x];
        x = he_ccw(&h, x);
      } while (x != xx && ne <= m->ve_ptr[v + 1] - m->ve_ptr[v]);
      if (ne != m->ve_ptr[v + 1] - m->ve_ptr[v] || nf != m->vf_ptr[v + 1] - m->vf_ptr[v]) { ok = 0; break; }
    }
  }
  free(h.vh); free(h.hto); free(h.hface); free(h.hnext); free(h.hprev); free(h.fh);
  return ok;
}

mo_mesh* mo_mesh_create(uint32_t V, uint32_t F, const float* xyz, const uint32_t* faces)
{
  mo_mesh* m = (mo_mesh*)calloc(1, sizeof(mo_mesh));
  m->V = V; m->F = F;
  m->xyz = (float*)malloc(sizeof(float) * 3 * (size_t)(V ? V : 1));
  memcpy(m->xyz, xyz, sizeof(float) * 3 * (size_t)V);
  m->fv = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (size_t)(F ? F : 1));
  memcpy(m->fv, faces, sizeof(uint32_t) * 3 * (size_t)F);
  m->fe = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (size_t)(F ? F : 1));

  /* edges in order of first appearance (CONVENTION) */
  size_t cap = 16;
  while (cap < (size_t)F * 6 + 16) cap <<= 1;
  uint64_t* hk = (uint64_t*)malloc(sizeof(uint64_t) * cap);
  uint32_t* hv = (uint32_t*)malloc(sizeof(uint32_t) * cap);
  memset(hv, 0xFF, sizeof(uint32_t) * cap);
  uint32_t* ev = (uint32_t*)malloc(sizeof(uint32_t) * 2 * ((size_t)F * 3 + 1));
  uint32_t E = 0;
  for (uint32_t f = 0; f < F; ++f) {
    for (int k = 0; k < 3; ++k) {
      const uint32_t a = faces[3 * f + k], b = faces[3 * f + (k + 1) % 3];
      const uint64_t key = edge_key(a, b);
      size_t h = hash64(key) & (cap - 1);
      while (hv[h] != NONE && hk[h] != key) h = (h + 1) & (cap - 1);
      if (hv[h] == NONE) {
        hk[h] = key; hv[h] = E;
        ev[2 * E] = a; ev[2 * E + 1] = b;
        ++E;
      }
      m->fe[3 * f + k] = hv[h];
    }
  }
  free(hk); free(hv);
  m->E = E;
  m->ev = (uint32_t*)realloc(ev, sizeof(uint32_t) * 2 * ((size_t)E + 1));

  m->ef = (uint32_t*)malloc(sizeof(uint32_t) * 2 * ((size_t)E + 1));
  memset(m->ef, 0xFF, sizeof(uint32_t) * 2 * ((size_t)E + 1));
  for (uint32_t f = 0; f < F; ++f)
    for (int k = 0; k < 3; ++k) {
      const uint32_t e = m->fe[3 * f + k];
      if (m->ef[2 * e] == NONE) m->ef[2 * e] = f;
      else if (m->ef[2 * e + 1] == NONE && m->ef[2 * e] != f) m->ef[2 * e + 1] = f;
    }

  /* vertex -> edges (getEdgesOfVertex), first in ascending edge id */
  m->ve_ptr = (uint32_t*)calloc((size_t)V + 2, sizeof(uint32_t));
  for (uint32_t e = 0; e < E; ++e) { m->ve_ptr[m->ev[2 * e] + 1]++; m->ve_ptr[m->ev[2 * e + 1] + 1]++; }
  for (uint32_t v = 0; v < V; ++v) m->ve_ptr[v + 1] += m->ve_ptr[v];
  m->ve = (uint32_t*)malloc(sizeof(uint32_t) * (2 * (size_t)E + 1));
  {
    uint32_t* fill = (uint32_t*)calloc((size_t)V + 1, sizeof(uint32_t));
    for (uint32_t e = 0; e < E; ++e)
      for (int s = 0; s < 2; ++s) {
        const uint32_t v = m->ev[2 * e + s];
        m->ve[m->ve_ptr[v] + fill[v]++] = e;
      }
    free(fill);
  }
  /* vertex -> faces (getFacesOfVertex), first in ascending face id */
  m->vf_ptr = (uint32_t*)calloc((size_t)V + 2, sizeof(uint32_t));
  for (uint32_t f = 0; f < F; ++f)
    for (int k = 0; k < 3; ++k) m->vf_ptr[faces[3 * f + k] + 1]++;
  for (uint32_t v = 0; v < V; ++v) m->vf_ptr[v + 1] += m->vf_ptr[v];
  m->vf = (uint32_t*)malloc(sizeof(uint32_t) * (3 * (size_t)F + 1));
  {
    uint32_t* fill = (uint32_t*)calloc((size_t)V + 1, sizeof(uint32_t));
    for (uint32_t f = 0; f < F; ++f)
      for (int k = 0; k < 3; ++k) {
        const uint32_t v = faces[3 * f + k];
        m->vf[m->vf_ptr[v] + fill[v]++] = f;
      }
    free(fill);
  }
  /* rows so far in ascending id; now into the half-edge circulator order of lvr2::PMPMesh */
  m->manifold = he_circulator_orders(m);
  return m;
}
int mo_mesh_is_manifold(const mo_mesh* m) { return m->manifold; }
void mo_mesh_vertex_faces(const mo_mesh* m, uint32_t* vf_ptr, uint32_t* vf)
{
  memcpy(vf_ptr, m->vf_ptr, sizeof(uint32_t) * ((size_t)m->V + 1));
  memcpy(vf, m->vf, sizeof(uint32_t) * 3 * (size_t)m->F);
}
void mo_mesh_vertex_edges(const mo_mesh* m, uint32_t* ve_ptr, uint32_t* ve)
{
  memcpy(ve_ptr, m->ve_ptr, sizeof(uint32_t) * ((size_t)m->V + 1));
  memcpy(ve, m->ve, sizeof(uint32_t) * 2 * (size_t)m->E);
}

void mo_mesh_destroy(mo_mesh* m)
{
  if (!m) return;
  free(m->xyz); free(m->fv); free(m->fe); free(m->ev); free(m->ef);
  free(m->ve_ptr); free(m->ve); free(m->vf_ptr); free(m->vf);
  free(m);
}

uint32_t mo_mesh_num_vertices(const mo_mesh* m) { return m->V; }
uint32_t mo_mesh_num_faces(const mo_mesh* m) { return m->F; }
uint32_t mo_mesh_num_edges(const mo_mesh* m) { return m->E; }
void mo_mesh_edges(const mo_mesh* m, uint32_t* edge_vtx) { memcpy(edge_vtx, m->ev, sizeof(uint32_t) * 2 * (size_t)m->E); }
void mo_mesh_face_edges(const mo_mesh* m, uint32_t* face_edges) { memcpy(face_edges, m->fe, sizeof(uint32_t) * 3 * (size_t)m->F); }

static vec3 vpos(const mo_mesh* m, uint32_t v) { return v3_load(m->xyz + 3 * (size_t)v); }

void mo_edge_distances(const mo_mesh* m, float* edge_dist)
{
  for (uint32_t e = 0; e < m->E; ++e)
    edge_dist[e] = v3_distance(vpos(m, m->ev[2 * e]), vpos(m, m->ev[2 * e + 1]));
}

void mo_face_normals(const mo_mesh* m, float* fn)
{
  for (uint32_t f = 0; f < m->F; ++f) {
    const vec3 a = vpos(m, m->fv[3 * f]), b = vpos(m, m->fv[3 * f + 1]), c = vpos(m, m->fv[3 * f + 2]);
    v3_store(fn + 3 * (size_t)f, v3_normalized(v3_cross(v3_sub(b, a), v3_sub(c, a))));
  }
}

void mo_vertex_normals(const mo_mesh* m, const float* fn, float* vn)
{
  for (uint32_t v = 0; v < m->V; ++v) {
    vec3 s = v3(0, 0, 0);
    for (uint32_t i = m->vf_ptr[v]; i < m->vf_ptr[v + 1]; ++i) s = v3_add(s, v3_load(fn + 3 * (size_t)m->vf[i]));
    v3_store(vn + 3 * (size_t)v, v3_normalized(s));
  }
}

/* mesh_map/src/mesh_map.cpp:517-561 */
void mo_compute_edge_weights(const mo_mesh* m, const float* edge_dist, const float* vertex_costs,
                             double edge_cost_factor, float* edge_weights)
{
  for (uint32_t e = 0; e < m->E; ++e) {
    const float v1cost = vertex_costs[m->ev[2 * e]];      /* :528 */
    const float v2cost = vertex_costs[m->ev[2 * e + 1]];  /* :529 */
    if (isinf(v1cost) || isinf(v2cost)) {                 /* :538 */
      edge_weights[e] = INFINITY;                         /* :541 */
    } else {
      const float vertex_dist = edge_dist[e];             /* :548 */
      /* :550  float*(float+float) promoted to double for "/ 2.0", stored as float */
      const float edge_cost = (float)((double)(vertex_dist * (v1cost + v2cost)) / 2.0);
      /* :552  float + double*float -> double -> float */
      edge_weights[e] = (float)((double)vertex_dist + edge_cost_factor * (double)edge_cost);
    }
  }
}

/* mesh_layers/src/steepness_layer.cpp:157-166 and :82-93 */
void mo_steepness(const mo_mesh* m, const float* vertex_normals, double threshold, float* steepness,
                  uint8_t* lethal)
{
  for (uint32_t v = 0; v < m->V; ++v) {
    /* :165  acos(float) -> the double overload is not selected for a float
     * argument under <cmath>; CONVENTION: acosf. */
    steepness[v] = acosf(vertex_normals[3 * (size_t)v + 2]);
    lethal[v] = (double)steepness[v] > threshold ? 1 : 0; /* :88 */
  }
}

/* ------------------------------------------------------------------------- */
/* lvr2::Meap<VertexHandle,float> emulation.  CONVENTION (lvr2 un-vendored):  */
/* array binary min-heap + key->position index; insert() = insert-or-update.  */
/* Which of several entries with EQUAL value popMin() returns is an internal  */
/* of lvr2 that the reference does not pin; we fix it: among equal values the */
/* smaller vertex id pops first, i.e. the heap is ordered by (value, id).     */
/* The device path implements the same total order (DESIGN.md "tie rule").    */
/* ------------------------------------------------------------------------- */
struct mo_meap {
  uint32_t n, cap;
  uint32_t* keys;  /* heap array: vertex id */
  float* vals;     /* heap array: value */
  uint32_t* pos;   /* vertex id -> heap index, NONE if absent */
};

mo_meap* mo_meap_create(uint32_t capacity)
{
  mo_meap* h = (mo_meap*)calloc(1, sizeof(mo_meap));
  h->cap = capacity;
  h->keys = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)capacity + 1));
  h->vals = (float*)malloc(sizeof(float) * ((size_t)capacity + 1));
  h->pos = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)capacity + 1));
  memset(h->pos, 0xFF, sizeof(uint32_t) * ((size_t)capacity + 1));
  return h;
}
void mo_meap_destroy(mo_meap* h) { if (h) { free(h->keys); free(h->vals); free(h->pos); free(h); } }
int mo_meap_empty(const mo_meap* h) { return h->n == 0; }

static void meap_swap(mo_meap* h, uint32_t i, uint32_t j)
{
  const uint32_t ki = h->keys[i], kj = h->keys[j];
  const float vi = h->vals[i], vj = h->vals[j];
  h->keys[i] = kj; h->vals[i] = vj; h->pos[kj] = i;
  h->keys[j] = ki; h->vals[j] = vi; h->pos[ki] = j;
}
/* 1 (default): equal values pop in ascending vertex id -- the tie rule the device path implements.
 * 0: values only, i.e. the plain array heap of lvr2::Meap as modelled in oracle/ref_build/stubs (which of
 *    several equal keys pops first then follows from the sift mechanics); used to compare tie-heavy inputs
 *    (all inflation seeds sit at 0) with oracle/_ref. */
static int g_meap_ties_by_id = 1;
void mo_set_heap_ties_by_id(int on) { g_meap_ties_by_id = on; }
static int meap_lt(const mo_meap* h, uint32_t i, uint32_t j)
{
  return h->vals[i] < h->vals[j] || (g_meap_ties_by_id && h->vals[i] == h->vals[j] && h->keys[i] < h->keys[j]);
}
static void meap_up(mo_meap* h, uint32_t i)
{
  while (i > 0) {
    const uint32_t p = (i - 1) / 2;
    if (!meap_lt(h, i, p)) break;
    meap_swap(h, i, p);
    i = p;
  }
}
static void meap_down(mo_meap* h, uint32_t i)
{
  for (;;) {
    const uint32_t l = 2 * i + 1, r = 2 * i + 2;
    const int lsm = l < h->n && meap_lt(h, l, i);
    const int rsm = r < h->n && meap_lt(h, r, i);
    if (!lsm && !rsm) break;
    uint32_t c;
    if (r >= h->n) c = l;
    else c = meap_lt(h, l, r) ? l : r;
    meap_swap(h, i, c);
    i = c;
  }
}
void mo_meap_insert(mo_meap* h, uint32_t key, float value)
{
  const uint32_t p = h->pos[key];
  if (p != NONE) {
    const float prev = h->vals[p];
    h->vals[p] = value;
    if (value < prev) meap_up(h, p);
    else if (value > prev) meap_down(h, p);
  } else {
    const uint32_t i = h->n++;
    h->keys[i] = key; h->vals[i] = value; h->pos[key] = i;
    meap_up(h, i);
  }
}
uint32_t mo_meap_pop_min(mo_meap* h, float* value)
{
  const uint32_t k = h->keys[0];
  if (value) *value = h->vals[0];
  h->pos[k] = NONE;
  h->n--;
  if (h->n > 0) {
    h->keys[0] = h->keys[h->n]; h->vals[0] = h->vals[h->n]; h->pos[h->keys[0]] = 0;
    meap_down(h, 0);
  }
  return k;
}

/* ------------------------------------------------------------------------- */
/* Dijkstra: dijkstra_mesh_planner/src/dijkstra_mesh_planner.cpp:217-398      */
/* ------------------------------------------------------------------------- */
uint32_t mo_dijkstra(const mo_mesh* m, const float* edge_weights, const float* vertex_costs,
                     uint8_t* invalid, uint32_t start_vertex /* wave seed */,
                     uint32_t goal_vertex /* robot */, double goal_dist_offset, double cost_limit,
                     float* distances, uint32_t* predecessors, uint32_t* path, uint32_t* path_len,
                     const volatile int* cancel, mo_stats* st)
{
  const double t0 = now_ms();
  mo_stats s; memset(&s, 0, sizeof(s)); s.goal_dist = INFINITY;
  const uint32_t V = m->V;
  *path_len = 0;                                            /* :248 path.clear() */
  /* :266-270 (also stands for the cleared maps of :249-250) */
  for (uint32_t v = 0; v < V; ++v) { distances[v] = INFINITY; predecessors[v] = v; }
  if (start_vertex >= V) { if (st) *st = s; return MO_INVALID_START; }  /* :240-243 */
  if (goal_vertex >= V) { if (st) *st = s; return MO_INVALID_GOAL; }
  if (goal_vertex == start_vertex) { if (st) *st = s; return MO_SUCCESS; } /* :252-255 */

  uint8_t* fixed = (uint8_t*)calloc((size_t)V + 1, 1);      /* :257 */
  mo_meap* pq = mo_meap_create(V);                          /* :272 */
  distances[start_vertex] = 0;                              /* :276 */
  mo_meap_insert(pq, start_vertex, 0);                      /* :277 */
  float goal_dist = INFINITY;                               /* :279 */
  const double t1 = now_ms();

  while (!mo_meap_empty(pq) && !(cancel && *cancel)) {      /* :287 */
    const uint32_t cur = mo_meap_pop_min(pq, NULL);         /* :289 */
    fixed[cur] = 1;                                         /* :290 */
    s.fixed_set_cnt++;
    if (cur == goal_vertex)                                 /* :293 */
      goal_dist = (float)((double)distances[cur] + goal_dist_offset); /* :296 */
    if (distances[cur] > goal_dist) continue;               /* :299 */
    if ((double)vertex_costs[cur] > cost_limit) continue;   /* :302 */
    s.expanded++;
    for (uint32_t i = m->ve_ptr[cur]; i < m->ve_ptr[cur + 1]; ++i) { /* :305-321 */
      const uint32_t eH = m->ve[i];
      const uint32_t vH = m->ev[2 * eH] == cur ? m->ev[2 * eH + 1] : m->ev[2 * eH]; /* :324-325 */
      s.edge_visits++;
      if (fixed[vH]) continue;                              /* :326 */
      if (invalid && invalid[vH]) continue;                 /* :328 */
      const float tmp_cost = distances[cur] + edge_weights[eH]; /* :331 */
      if (tmp_cost < distances[vH]) {                       /* :332 */
        distances[vH] = tmp_cost;
        mo_meap_insert(pq, vH, tmp_cost);                   /* :335 */
        predecessors[vH] = cur;                             /* :336 */
        s.relaxations++;
      }
    }
  }
  const double t2 = now_ms();
  s.goal_dist = goal_dist;
  s.t_init_ms = t1 - t0; s.t_propagation_ms = t2 - t1;
  mo_meap_destroy(pq); free(fixed);
  if (cancel && *cancel) { if (st) *st = s; return MO_CANCELED; }       /* :350-354 */
  if (goal_vertex == predecessors[goal_vertex]) { if (st) *st = s; return MO_NO_PATH_FOUND; } /* :358 */

  /* :367-373  path.push_front(pred...) until the seed is reached.  We emit the
   * list front-to-back, i.e. seed first. */
  uint32_t n = 0;
  for (uint32_t vH = goal_vertex; vH != start_vertex;) { vH = predecessors[vH]; n++; }
  *path_len = n;
  {
    uint32_t vH = goal_vertex, i = n;
    while (vH != start_vertex) { vH = predecessors[vH]; path[--i] = vH; }
  }
  s.t_backtrack_ms = now_ms() - t2;
  if (st) *st = s;
  return MO_SUCCESS;
}

/* Deterministic predecessor rule used by the device path. */
void mo_dijkstra_pred_rule(const mo_mesh* m, const float* edge_weights, const float* vertex_costs,
                           const uint8_t* invalid, uint32_t seed_vertex, float goal_dist,
                           double cost_limit, const float* dist, uint32_t* pred)
{
  for (uint32_t v = 0; v < m->V; ++v) {
    pred[v] = v;
    if (v == seed_vertex || !(dist[v] < INFINITY) || (invalid && invalid[v])) continue;
    float best_du = INFINITY; uint32_t best_u = NONE;
    for (uint32_t i = m->ve_ptr[v]; i < m->ve_ptr[v + 1]; ++i) {
      const uint32_t e = m->ve[i];
      const uint32_t u = m->ev[2 * e] == v ? m->ev[2 * e + 1] : m->ev[2 * e];
      const float du = dist[u];
      if (!(du < INFINITY)) continue;
      if (du > goal_dist) continue;                       /* never expanded */
      if ((double)vertex_costs[u] > cost_limit) continue; /* never expanded */
      const float sum = du + edge_weights[e];
      if (sum != dist[v]) continue;
      if (du < best_du || (du == best_du && u < best_u)) { best_du = du; best_u = u; }
    }
    if (best_u != NONE) pred[v] = best_u;
  }
}

/* dijkstra_mesh_planner.cpp:189-209 */
void mo_dijkstra_vector_map(const mo_mesh* m, const uint32_t* pred, float* vecmap)
{
  for (uint32_t v3i = 0; v3i < m->V; ++v3i) {
    const uint32_t v1 = pred[v3i];
    if (v1 == v3i) continue;                                 /* :197 */
    const vec3 dir = v3_sub(vpos(m, v1), vpos(m, v3i));      /* :204 */
    v3_store(vecmap + 3 * (size_t)v3i, v3_normalized(dir));  /* :206 */
  }
}

/* ------------------------------------------------------------------------- */
/* CVP: cvp_mesh_planner/src/cvp_mesh_planner.cpp                             */
/* ------------------------------------------------------------------------- */

/* :369-556 on plain numbers.  Returns 1 when the reference returns true. */
int mo_cvp_update_scalar(float u1f, float u2f, float u3f, float af, float bf, float cf,
                         float* u3_out, int* pred_sel, float* direction)
{
  const double u1 = u1f, u2 = u2f, u3 = u3f;                 /* :376-378 */
  const double c = cf, c_sq = c * c;                         /* :381-382 */
  const double b = bf, b_sq = b * b;                         /* :385-386 */
  const double a = af, a_sq = a * a;                         /* :389-390 */
  const double u1_sq = u1 * u1, u2_sq = u2 * u2;             /* :392-393 */
  const double sx = (c_sq + u1_sq - u2_sq) / (2 * c);        /* :395 */
  const double sy = -sqrt(fmax(u1_sq - sx * sx, 0.0));       /* :396 */
  const double p = (b_sq + c_sq - a_sq) / (2 * c);           /* :398 */
  const double hc = sqrt(fmax(b_sq - p * p, 0.0));           /* :399 */
  const double dy = hc - sy, dx = p - sx;                    /* :401-402 */
  const double u3tmp_sq = dx * dx + dy * dy;                 /* :404 */
  double u3tmp = sqrt(u3tmp_sq);                             /* :405 */
  if (u3tmp < u3) {                                          /* :411 */
    const double t0a = (a_sq + b_sq - c_sq) / (2 * a * b);             /* :413 */
    const double t1a = (u3tmp_sq + b_sq - u1_sq) / (2 * u3tmp * b);    /* :414 */
    const double t2a = (a_sq + u3tmp_sq - u2_sq) / (2 * a * u3tmp);    /* :415 */
    if (fabs(t1a) > 1) {                                     /* :418 */
      u3tmp = u1 + b;
      if (u3tmp < u3) { *u3_out = (float)u3tmp; *pred_sel = 1; *direction = 0; return 1; }
      return 0;
    } else if (fabs(t2a) > 1) {                              /* :437 */
      u3tmp = u2 + a;
      if (u3tmp < u3) { *u3_out = (float)u3tmp; *pred_sel = 2; *direction = 0; return 1; }
      return 0;
    }
    const double theta0 = acos(t0a), theta1 = acos(t1a), theta2 = acos(t2a); /* :456-458 */
    if (theta1 < theta0 && theta2 < theta0) {                /* :493 */
      *u3_out = (float)u3tmp;                                /* :497 */
      if (theta1 < theta2) { *pred_sel = 1; *direction = (float)theta1; }   /* :498-501 */
      else { *pred_sel = 2; *direction = (float)(-theta2); }               /* :507-510 */
      return 1;
    } else if (theta1 < theta2) {                            /* :518 */
      u3tmp = u1 + b;
      if (u3tmp < u3) { *u3_out = (float)u3tmp; *pred_sel = 1; *direction = 0; return 1; }
      return 0;
    } else {                                                 /* :536 */
      u3tmp = u2 + a;
      if (u3tmp < u3) { *u3_out = (float)u3tmp; *pred_sel = 2; *direction = 0; return 1; }
      return 0;
    }
  }
  return 0;
}

/* waveFrontUpdate bound to a mesh face: (v1,v2,v3) is a cyclic rotation of
 * face f with v3 at corner k3 (cvp_mesh_planner.cpp:811,834,857). */
static int cvp_update(const mo_mesh* m, const float* edge_weights, float* distances,
                      uint32_t* predecessors, float* direction, uint32_t* cutface, uint32_t f,
                      int k3)
{
  const int k1 = (k3 + 1) % 3, k2 = (k3 + 2) % 3;
  const uint32_t v1 = m->fv[3 * f + k1], v2 = m->fv[3 * f + k2], v3i = m->fv[3 * f + k3];
  /* fe[k] joins fv[k], fv[k+1]:  c=(v1,v2)=fe[k1]; b=(v1,v3)=fe[k3]; a=(v2,v3)=fe[k2] */
  const float c = edge_weights[m->fe[3 * f + k1]];
  const float b = edge_weights[m->fe[3 * f + k3]];
  const float a = edge_weights[m->fe[3 * f + k2]];
  float u3n, dir; int sel;
  if (!mo_cvp_update_scalar(distances[v1], distances[v2], distances[v3i], a, b, c, &u3n, &sel, &dir))
    return 0;
  cutface[v3i] = f;                         /* cutting_faces_.insert(v3, fH) */
  predecessors[v3i] = sel == 1 ? v1 : v2;
  distances[v3i] = u3n;
  direction[v3i] = dir;
  return 1;
}

/* CVPMeshPlanner::computeVectorMap, cvp_mesh_planner.cpp:204-239 */
static void cvp_vector_map(const mo_mesh* m, const float* vertex_normals, const uint32_t* pred,
                           const float* direction, const uint32_t* cutface, float* vecmap,
                           uint8_t* has_vec)
{
  for (uint32_t v3i = 0; v3i < m->V; ++v3i) {
    const uint32_t v1 = pred[v3i];
    if (v1 == v3i) continue;                 /* :218 */
    if (cutface[v3i] == NONE) continue;      /* :222-225 */
    const vec3 d = v3_sub(vpos(m, v1), vpos(m, v3i));
    const vec3 r = v3_rotated(d, v3_load(vertex_normals + 3 * (size_t)v3i), direction[v3i]); /* :234 */
    v3_store(vecmap + 3 * (size_t)v3i, v3_normalized(r));                                    /* :236 */
    has_vec[v3i] = 1;
  }
}

uint32_t mo_cvp_propagate(const mo_mesh* m, const float* edge_weights, const float* vertex_costs,
                          uint8_t* invalid, const float* vertex_normals, const float seed_pos[3],
                          uint32_t start_face /* wave seed face */, uint32_t goal_face /* robot */,
                          double goal_dist_offset, double cost_limit, float* distances,
                          uint32_t* predecessors, float* direction, uint32_t* cutface, float* vecmap,
                          uint8_t* has_vec, const volatile int* cancel, mo_stats* st)
{
  const double t0 = now_ms();
  mo_stats s; memset(&s, 0, sizeof(s)); s.goal_dist = INFINITY;
  const uint32_t V = m->V;
  if (start_face >= m->F) { if (st) *st = s; return MO_INVALID_START; }  /* :681-685 */
  if (goal_face >= m->F) { if (st) *st = s; return MO_INVALID_GOAL; }    /* :686-690 */
  const vec3 start = v3_load(seed_pos);

  uint8_t* fixed = (uint8_t*)calloc((size_t)V + 1, 1);        /* :702 */
  memset(vecmap, 0, sizeof(float) * 3 * (size_t)V);           /* :705 vector_map_.clear() */
  memset(has_vec, 0, (size_t)V);
  for (uint32_t v = 0; v < V; ++v) { distances[v] = INFINITY; predecessors[v] = v; } /* :710-714 */

  mo_meap* pq = mo_meap_create(V);                            /* :716 */
  for (int k = 0; k < 3; ++k) {                               /* :719-728 */
    const uint32_t vH = m->fv[3 * start_face + k];
    const vec3 diff = v3_sub(start, vpos(m, vH));
    const float dist = v3_length(diff);
    distances[vH] = dist;
    v3_store(vecmap + 3 * (size_t)vH, diff); has_vec[vH] = 1;
    cutface[vH] = start_face;
    fixed[vH] = 1;
    mo_meap_insert(pq, vH, dist);
  }
  const uint32_t g0 = m->fv[3 * goal_face], g1 = m->fv[3 * goal_face + 1], g2 = m->fv[3 * goal_face + 2]; /* :730 */
  float goal_dist = INFINITY;                                 /* :738 */
  const double t1 = now_ms();

  while (!mo_meap_empty(pq) && !(cancel && *cancel)) {        /* :747 */
    const uint32_t cur = mo_meap_pop_min(pq, NULL);           /* :749 */
    fixed[cur] = 1;                                           /* :751 */
    s.fixed_set_cnt++;
    if (distances[cur] > goal_dist) continue;                 /* :754 */
    if ((double)vertex_costs[cur] >= cost_limit) continue;    /* :757 */
    if (invalid && invalid[cur]) continue;                    /* :760 */
    if (cur == g0 || cur == g1 || cur == g2) {                /* :763 */
      if (goal_dist == INFINITY && fixed[g0] && fixed[g1] && fixed[g2])   /* :765-766 */
        goal_dist = (float)((double)distances[cur] + goal_dist_offset);   /* :769 */
    }
    s.expanded++;
    for (uint32_t i = m->vf_ptr[cur]; i < m->vf_ptr[cur + 1]; ++i) {      /* :775-778 */
      const uint32_t fh = m->vf[i];
      const uint32_t a = m->fv[3 * fh], b = m->fv[3 * fh + 1], c = m->fv[3 * fh + 2]; /* :780-783 */
      s.edge_visits++;
      if (invalid && (invalid[a] || invalid[b] || invalid[c])) continue;  /* :785 */
      int k3;
      uint32_t free_v;
      if (fixed[a] && fixed[b] && fixed[c]) continue;                     /* :790 */
      else if (fixed[a] && fixed[b] && !fixed[c]) { k3 = 2; free_v = c; } /* :797  (a,b,c) */
      else if (fixed[a] && !fixed[b] && fixed[c]) { k3 = 1; free_v = b; } /* :820  (c,a,b) */
      else if (!fixed[a] && fixed[b] && fixed[c]) { k3 = 0; free_v = a; } /* :843  (b,c,a) */
      else continue;                                                      /* :866-870 */
      if ((double)vertex_costs[free_v] >= cost_limit) continue;           /* :802,825,848 */
      if (cvp_update(m, edge_weights, distances, predecessors, direction, cutface, fh, k3)) {
        mo_meap_insert(pq, free_v, distances[free_v]);                    /* :814,837,860 */
        s.relaxations++;
      }
    }
  }
  const double t2 = now_ms();
  s.goal_dist = goal_dist;
  s.t_init_ms = t1 - t0; s.t_propagation_ms = t2 - t1;
  mo_meap_destroy(pq); free(fixed);
  if (cancel && *cancel) { if (st) *st = s; return MO_CANCELED; }         /* :888-892 */

  cvp_vector_map(m, vertex_normals, predecessors, direction, cutface, vecmap, has_vec); /* :897 */
  s.t_backtrack_ms = now_ms() - t2;
  if (st) *st = s;

  /* :902-918 */
  int any_pred = 0;
  if (g0 != predecessors[g0] || g1 != predecessors[g1] || g2 != predecessors[g2]) any_pred = 1;
  if (!any_pred && goal_face != start_face) return MO_NO_PATH_FOUND;
  return MO_SUCCESS;
}

/* mesh_map/src/util.cpp:320-347 */
int mo_projected_barycentric(const float pp[3], const float pa[3], const float pb[3],
                             const float pc[3], float bary[3], float* dist)
{
  const vec3 p = v3_load(pp), a = v3_load(pa), b = v3_load(pb), c = v3_load(pc);
  const vec3 u = v3_sub(b, a), v = v3_sub(c, a), w = v3_sub(p, a);
  const vec3 n = v3_cross(u, v);
  const float oneOver4ASquared = (float)(1.0 / (double)v3_dot(n, n));    /* :333 */
  const float gamma = v3_dot(v3_cross(u, w), n) * oneOver4ASquared;      /* :335 */
  const float beta = v3_dot(v3_cross(w, v), n) * oneOver4ASquared;       /* :337 */
  const float alpha = 1 - gamma - beta;                                  /* :338 */
  bary[0] = alpha; bary[1] = beta; bary[2] = gamma;
  *dist = v3_dot(n, w) / v3_length(n);                                   /* :341 */
  const float EPSILON = 0.01f;                                           /* :343 */
  return ((0 - EPSILON <= alpha) && (alpha <= 1 + EPSILON) && (0 - EPSILON <= beta) &&
          (beta <= 1 + EPSILON) && (0 - EPSILON <= gamma) && (gamma <= 1 + EPSILON));
}

static int face_bary(const mo_mesh* m, vec3 p, uint32_t f, float bary[3], float* dist)
{
  float pp[3]; v3_store(pp, p);
  return mo_projected_barycentric(pp, m->xyz + 3 * (size_t)m->fv[3 * f], m->xyz + 3 * (size_t)m->fv[3 * f + 1],
                                  m->xyz + 3 * (size_t)m->fv[3 * f + 2], bary, dist);
}

/* mesh_map.cpp:1161-1174 (nanoflann 1-NN, L2_Simple metric: sum of squared
 * float differences); brute force, first minimum wins. */
uint32_t mo_nearest_vertex(const mo_mesh* m, const float p[3])
{
  uint32_t best = NONE; float bd = INFINITY;
  for (uint32_t v = 0; v < m->V; ++v) {
    const float dx = p[0] - m->xyz[3 * (size_t)v], dy = p[1] - m->xyz[3 * (size_t)v + 1], dz = p[2] - m->xyz[3 * (size_t)v + 2];
    const float d = dx * dx + dy * dy + dz * dz;
    if (d < bd) { bd = d; best = v; }
  }
  return best;
}

/* mesh_map.cpp:1120-1159 */
uint32_t mo_containing_face(const mo_mesh* m, const float p[3], float bary_out[3])
{
  const uint32_t vH = mo_nearest_vertex(m, p);
  if (vH == NONE) return NONE;
  float lowest = FLT_MAX;                                                /* :1131 */
  uint32_t best = NONE;
  for (uint32_t i = m->vf_ptr[vH]; i < m->vf_ptr[vH + 1]; ++i) {         /* :1135 */
    const uint32_t f = m->vf[i];
    float bary[3], dist = 0;
    const int inside = face_bary(m, v3_load(p), f, bary, &dist);         /* :1140 */
    if (inside && dist < lowest) {                                       /* :1142 (signed) */
      lowest = dist; best = f;
      if (bary_out) { bary_out[0] = bary[0]; bary_out[1] = bary[1]; bary_out[2] = bary[2]; }
    }
  }
  return best;
}

/* InflationLayer::vectorAt(handles, bary), inflation_layer.cpp:493-521 */
static vec3 inflation_vector_at(const mo_inflation_field* L, const uint32_t vs[3], const float bary[3])
{
  if (!L->repulsive_field) return v3(0, 0, 0);
  /* linearCombineBarycentricCoords(vertices, distances_, bary) util.h:195-203 */
  const float distance = L->distances[vs[0]] * bary[0] + L->distances[vs[1]] * bary[1] + L->distances[vs[2]] * bary[2];
  if ((double)distance > L->cfg.inflation_radius) return v3(0, 0, 0);   /* :501 */
  const vec3 comb = v3_add(v3_add(v3_scale(v3_load(L->vecmap + 3 * (size_t)vs[0]), bary[0]),
                                  v3_scale(v3_load(L->vecmap + 3 * (size_t)vs[1]), bary[1])),
                           v3_scale(v3_load(L->vecmap + 3 * (size_t)vs[2]), bary[2]));
  if ((double)distance > L->cfg.inscribed_radius) {                      /* :505 */
    /* :507-510.  sqrt(float) - double ... evaluated in double, stored float */
    const float alpha = (float)(((double)sqrtf(distance) - L->cfg.inscribed_radius) /
                                (L->cfg.inflation_radius - L->cfg.inscribed_radius) * M_PI);
    /* :509-510  vec * inscribed_value * (cos(alpha) + 1) / 2.0 evaluates left to right on
     * lvr2::BaseVector<float>, whose operator* and operator/ take a float: three float
     * vector operations, each scalar narrowed first (checked against oracle/_ref) */
    return v3_div(v3_scale(v3_scale(comb, (float)L->cfg.inscribed_value), cosf(alpha) + 1), 2.0f);
  }
  if (distance > 0) return v3_scale(comb, (float)L->cfg.inscribed_value); /* :514-517 */
  return v3_scale(comb, (float)L->cfg.lethal_value);                      /* :520 */
}

void mo_inflation_vector_at(const mo_inflation_field* L, const uint32_t vs[3], const float bary[3], float out[3])
{
  v3_store(out, inflation_vector_at(L, vs, bary));
}

/* MeshMap::searchNeighbourFaces, mesh_map.cpp:999-1068.  Returns face or NONE. */
static uint32_t search_neighbour_faces(const mo_mesh* m, vec3 pos, uint32_t face, float max_radius,
                                       float max_dist, float bary_out[3])
{
  uint32_t* list = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)m->F + 1));
  uint8_t* in_list = (uint8_t*)calloc((size_t)m->F + 1, 1);
  uint32_t n = 0, it = 0, result = NONE;
  list[n++] = face; in_list[face] = 1;                                   /* :1005,1024 */
  vec3 center = v3(0, 0, 0);
  for (int k = 0; k < 3; ++k) center = v3_add(center, vpos(m, m->fv[3 * face + k])); /* :1010-1013 */
  center = v3_div(center, 3);                                            /* :1014 */
  float vertex_center_max = 0;
  for (int k = 0; k < 3; ++k) vertex_center_max = fmaxf(vertex_center_max, v3_distance(vpos(m, m->fv[3 * face + k]), center)); /* :1017-1020 */
  const float ext_radius = max_radius + vertex_center_max;               /* :1022 */
  const float max_radius_sq = ext_radius * ext_radius;                   /* :1023 */
  while (it < n) {                                                       /* :1031 */
    const uint32_t f = list[it];
    float bary[3], dist;
    if (face_bary(m, pos, f, bary, &dist) && fabsf(dist) < max_dist) {   /* :1035 */
      bary_out[0] = bary[0]; bary_out[1] = bary[1]; bary_out[2] = bary[2];
      result = f;
      break;
    }
    for (int k = 0; k < 3; ++k) {                                        /* :1042 */
      const uint32_t vertex = m->fv[3 * f + k];
      if (v3_distance2(center, vpos(m, vertex)) < max_radius_sq) {       /* :1044 */
        for (uint32_t i = m->vf_ptr[vertex]; i < m->vf_ptr[vertex + 1]; ++i) { /* :1048-1049 */
          const uint32_t nn = m->vf[i];
          if (!in_list[nn]) { list[n++] = nn; in_list[nn] = 1; }        /* :1051-1055 */
        }
      }
    }
    ++it;                                                                /* :1063 */
  }
  free(list); free(in_list);
  return result;
}

/* MeshMap::meshAhead mesh_map.cpp:1070-1108 (+directionAtPosition :625-650) */
static int mesh_ahead(const mo_mesh* m, const float* vecmap, const uint8_t* has_vec,
                      const mo_inflation_field* infl, vec3* pos, uint32_t* face, float step_size)
{
  float bary[3], dist;
  if (face_bary(m, *pos, *face, bary, &dist)) {                          /* :1075 */
  } else {
    const uint32_t nf = search_neighbour_faces(m, *pos, *face, step_size, 0.4f, bary); /* :1079 */
    if (nf == NONE) return 0;                                            /* :1090-1093 */
    *face = nf;
    /* :1087 linearCombineBarycentricCoords(vertices, bary) util.h:178-184 */
    *pos = v3_add(v3_add(v3_scale(vpos(m, m->fv[3 * nf]), bary[0]), v3_scale(vpos(m, m->fv[3 * nf + 1]), bary[1])),
                  v3_scale(vpos(m, m->fv[3 * nf + 2]), bary[2]));
  }
  const uint32_t* vs = m->fv + 3 * (size_t)*face;
  /* directionAtPosition :625-650 */
  if (!(has_vec[vs[0]] || has_vec[vs[1]] || has_vec[vs[2]])) return 0;
  vec3 vec = v3(0, 0, 0);
  for (int k = 0; k < 3; ++k)
    if (has_vec[vs[k]]) vec = v3_add(vec, v3_scale(v3_load(vecmap + 3 * (size_t)vs[k]), bary[k]));
  if (!(isfinite(vec.x) && isfinite(vec.y) && isfinite(vec.z))) return 0;
  vec3 dir = v3_normalized(vec);                                         /* :1096 */
  if (infl) dir = v3_add(dir, inflation_vector_at(infl, vs, bary));      /* :1099-1102 */
  dir = v3_normalized(dir);                                              /* :1103 */
  *pos = v3_add(*pos, v3_scale(dir, step_size));                         /* :1104 */
  return 1;
}

/* cvp_mesh_planner.cpp:920-951 */
uint32_t mo_cvp_backtrack(const mo_mesh* m, const float* vecmap, const uint8_t* has_vec,
                          const mo_inflation_field* infl, const float seed_pos[3], uint32_t seed_face,
                          const float target_pos[3], uint32_t target_face, double step_width,
                          uint32_t cap, float* path_pos, uint32_t* path_face, uint32_t* path_len)
{
  /* the reference push_front()s; we collect back-to-front and reverse at the end */
  const vec3 start = v3_load(seed_pos);
  uint32_t face = target_face;                                           /* :922 */
  vec3 pos = v3_load(target_pos);                                        /* :923 */
  uint32_t n = 0, code = MO_SUCCESS;
  if (n < cap) { v3_store(path_pos + 3 * (size_t)n, pos); path_face[n] = face; } n++;   /* :924 */
  while ((double)v3_distance2(pos, start) > step_width) {                /* :927 (squared vs width quirk) */
    if (mesh_ahead(m, vecmap, has_vec, infl, &pos, &face, (float)step_width)) {         /* :933 */
      if (n < cap) { v3_store(path_pos + 3 * (size_t)n, pos); path_face[n] = face; } n++; /* :935 */
      if (n >= cap) { code = MO_NO_PATH_FOUND; break; } /* oracle guard against endless loops */
    } else { code = MO_NO_PATH_FOUND; break; }                           /* :937-942 */
  }
  if (code == MO_SUCCESS) { if (n < cap) { v3_store(path_pos + 3 * (size_t)n, start); path_face[n] = seed_face; } n++; } /* :951 */
  if (n > cap) n = cap;
  /* reverse into reference list order (front = seed) */
  for (uint32_t i = 0; i < n / 2; ++i) {
    const uint32_t j = n - 1 - i;
    for (int k = 0; k < 3; ++k) { float t = path_pos[3 * i + k]; path_pos[3 * i + k] = path_pos[3 * j + k]; path_pos[3 * j + k] = t; }
    uint32_t tf = path_face[i]; path_face[i] = path_face[j]; path_face[j] = tf;
  }
  *path_len = n;
  return code;
}

/* ------------------------------------------------------------------------- */
/* poses: mesh_map/src/util.cpp:267-298                                       */
/* ------------------------------------------------------------------------- */
/* tf2::Matrix3x3::getRotation (tf2 un-vendored): published Bullet/tf2
 * matrix -> quaternion conversion, doubles; followed by normalize (:278). */
static void basis_to_quat(const double mm[3][3], double q[4])
{
  const double trace = mm[0][0] + mm[1][1] + mm[2][2];
  double temp[4];
  if (trace > 0.0) {
    double s = sqrt(trace + 1.0);
    temp[3] = s * 0.5; s = 0.5 / s;
    temp[0] = (mm[2][1] - mm[1][2]) * s;
    temp[1] = (mm[0][2] - mm[2][0]) * s;
    temp[2] = (mm[1][0] - mm[0][1]) * s;
  } else {
    const int i = mm[0][0] < mm[1][1] ? (mm[1][1] < mm[2][2] ? 2 : 1) : (mm[0][0] < mm[2][2] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = sqrt(mm[i][i] - mm[j][j] - mm[k][k] + 1.0);
    temp[i] = s * 0.5; s = 0.5 / s;
    temp[3] = (mm[k][j] - mm[j][k]) * s;
    temp[j] = (mm[j][i] + mm[i][j]) * s;
    temp[k] = (mm[k][i] + mm[i][k]) * s;
  }
  const double len = sqrt(temp[0] * temp[0] + temp[1] * temp[1] + temp[2] * temp[2] + temp[3] * temp[3]);
  for (int t = 0; t < 4; ++t) q[t] = temp[t] / len;
}

float mo_pose_from_position(const float current[3], const float next[3], const float normal[3],
                            double pose[7])
{
  const vec3 cur = v3_load(current), nrm = v3_load(normal);
  const vec3 direction = v3_sub(v3_load(next), cur);           /* :295 */
  const float cost = v3_length(direction);                     /* :296 */
  const vec3 ez = v3_normalized(nrm);                          /* :269 */
  const vec3 ey = v3_normalized(v3_cross(nrm, direction));     /* :270 */
  const vec3 ex = v3_normalized(v3_cross(ey, nrm));            /* :271 */
  const double basis[3][3] = { { ex.x, ey.x, ez.x }, { ex.y, ey.y, ez.y }, { ex.z, ey.z, ez.z } }; /* :273 */
  double q[4];
  basis_to_quat(basis, q);                                     /* :278-279 */
  pose[0] = cur.x; pose[1] = cur.y; pose[2] = cur.z;           /* :275,280 */
  pose[3] = q[0]; pose[4] = q[1]; pose[5] = q[2]; pose[6] = q[3];
  return cost;
}

/* dijkstra_mesh_planner.cpp:83-116.  `path` is dijkstra()'s list (seed first);
 * makePlan reverses it (:83) and walks from the robot position (start_vec) */
uint32_t mo_dijkstra_poses(const mo_mesh* m, const float* vertex_normals, const uint32_t* path,
                           uint32_t path_len, const float robot_pos[3], const float goal_pos[3],
                           double* poses, double* cost_out)
{
  double cost = 0;                                             /* :89 */
  uint32_t n = 0;
  if (path_len > 0) {                                          /* :90 */
    float vec[3] = { robot_pos[0], robot_pos[1], robot_pos[2] };            /* :92 */
    float normal[3];
    memcpy(normal, vertex_normals + 3 * (size_t)path[path_len - 1], sizeof(normal)); /* :94 path.front() after reverse */
    for (uint32_t i = path_len; i-- > 0;) {                    /* :100 */
      const uint32_t vH = path[i];
      const float* next = m->xyz + 3 * (size_t)vH;             /* :104 */
      const float dir_length = mo_pose_from_position(vec, next, normal, poses + 7 * (size_t)n); /* :106 */
      cost += dir_length;                                      /* :107 */
      memcpy(vec, next, sizeof(vec));                          /* :108 */
      memcpy(normal, vertex_normals + 3 * (size_t)vH, sizeof(normal));      /* :109 */
      n++;
    }
    const float dir_length = mo_pose_from_position(vec, goal_pos, normal, poses + 7 * (size_t)n); /* :113 */
    cost += dir_length;                                        /* :114 */
    n++;
  }
  *cost_out = cost;
  return n;
}

/* cvp_mesh_planner.cpp:93-124.  path in waveFrontPropagation list order (seed
 * first); makePlan reverses it (:93) -> robot first. */
uint32_t mo_cvp_poses(const mo_mesh* m, const float* face_normals, const float* path_pos,
                      const uint32_t* path_face, uint32_t path_len, const double goal_pose[7],
                      double* poses, double* cost_out)
{
  (void)m;
  double cost = 0;                                             /* :99 */
  uint32_t n = 0;
  if (path_len > 0) {                                          /* :101 */
    float vec[3];
    memcpy(vec, path_pos + 3 * (size_t)(path_len - 1), sizeof(vec));        /* :103 */
    uint32_t fH = path_face[path_len - 1];                     /* :104 */
    for (uint32_t i = path_len - 1; i-- > 0;) {                /* :108 */
      const float* next = path_pos + 3 * (size_t)i;
      const float dir_length = mo_pose_from_position(vec, next, face_normals + 3 * (size_t)fH, poses + 7 * (size_t)n); /* :112 */
      cost += dir_length;                                      /* :113 */
      memcpy(vec, next, sizeof(vec));                          /* :114 */
      fH = path_face[i];                                       /* :115 */
      n++;
    }
    memcpy(poses + 7 * (size_t)n, goal_pose, sizeof(double) * 7);           /* :119-123 */
    n++;
  }
  *cost_out = cost;
  return n;
}

/* ------------------------------------------------------------------------- */
/* Inflation layer: mesh_layers/src/inflation_layer.cpp                       */
/* ------------------------------------------------------------------------- */
#define MESH_LAYERS_EPSILON 1e-9f  /* mesh_layers::EPSILON, mesh_layers/include/mesh_layers/inflation_layer.h:48 (`const float EPSILON = 1e-9;`) */

/* :181-234 */
float mo_inflation_sethian(float d1, float d2, float a, float b, float dot, float F)
{
  float t = INFINITY;
  const float r_cos_angle = dot;
  const float r_sin_angle = sqrtf(1 - dot * dot);              /* :189 */
  const float u = d2 - d1;                                     /* :191 */
  const float f2 = a * a + b * b - 2 * a * b * r_cos_angle;    /* :193 */
  const float f1 = b * u * (a * r_cos_angle - b);              /* :194 */
  const float f0 = b * b * (u * u - F * F * a * a * r_sin_angle); /* :195 */
  const float delta = f1 * f1 - f0 * f2;                       /* :197 */
  if (delta >= 0) {
    if (fabsf(f2) > MESH_LAYERS_EPSILON) {                     /* :201 */
      t = (-f1 - sqrtf(delta)) / f2;                           /* :203 */
      if (t < u || b * (t - u) / t < a * r_cos_angle || a / r_cos_angle < b * (t - u) / 2) { /* :204 */
        t = (-f1 + sqrtf(delta)) / f2;                         /* :206 */
      } else {
        if (f1 != 0) t = -f0 / f1;                             /* :210-213 */
        else t = -INFINITY;                                    /* :216 */
      }
    }
  } else {
    t = -INFINITY;                                             /* :223 */
  }
  if (u < t && a * r_cos_angle < b * (t - u) / t && b * (t - u) / t < a / r_cos_angle) /* :226 */
    return t + d1;
  return fminf(b * F + d1, a * F + d2);                        /* :232 */
}

/* :315-339 */
float mo_inflation_fading(const mo_inflation_cfg* cfg, float distance)
{
  if ((double)distance > cfg->inflation_radius) return 0;      /* :317-320 */
  if ((double)distance > cfg->inscribed_radius) {              /* :323 */
    const float factor = (float)exp(-1.0 * cfg->cost_scaling_factor * ((double)distance - cfg->inscribed_radius)); /* :326 */
    const float cost = (float)(cfg->inscribed_value * (double)factor);   /* :327 */
    return cost;
  }
  if (distance > 0) return (float)cfg->inscribed_value;        /* :332-335 */
  return (float)cfg->lethal_value;                             /* :338 */
}

static uint32_t edge_between(const mo_mesh* m, uint32_t x, uint32_t y)
{
  for (uint32_t i = m->ve_ptr[x]; i < m->ve_ptr[x + 1]; ++i) {
    const uint32_t e = m->ve[i];
    if (m->ev[2 * e] == y || m->ev[2 * e + 1] == y) return e;
  }
  return NONE;
}

/* :236-313 */
int mo_inflation_wavefront_update(const mo_mesh* m, float* distances, float* vecmap, float max_distance,
                                  const float* edge_weights, uint32_t v1h, uint32_t v2h, uint32_t v3h)
{
  const float u1 = distances[v1h], u2 = distances[v2h], u3 = distances[v3h]; /* :248-250 (inf = absent) */
  if (u3 == 0) return 0;                                       /* :252 */
  const uint32_t e12 = edge_between(m, v1h, v2h), e13 = edge_between(m, v1h, v3h), e23 = edge_between(m, v2h, v3h);
  const float c = edge_weights[e12], c_sq = c * c;             /* :259-260 */
  const float b = edge_weights[e13], b_sq = b * b;             /* :262-263 */
  const float a = edge_weights[e23], a_sq = a * a;             /* :265-266 */
  const float dot = (a_sq + b_sq - c_sq) / (2 * a * b);        /* :268 */
  const float u3tmp = mo_inflation_sethian(u1, u2, a, b, dot, 1.0f); /* :269 */
  if (!isfinite(u3tmp)) return 0;                              /* :271 */
  const float d31 = u3tmp - u1, d32 = u3tmp - u2;              /* :274-275 */
  if (u1 == 0 && u2 == 0 && vecmap) {                          /* :277 */
    const vec3 p1 = vpos(m, v1h), p2 = vpos(m, v2h), p3 = vpos(m, v3h);
    const vec3 dir = v3_normalized(v3_add(v3_sub(p3, p2), v3_sub(p3, p1))); /* :282 */
    const uint32_t hs[3] = { v1h, v2h, v3h };
    for (int k = 0; k < 3; ++k)                                /* :293-295 */
      v3_store(vecmap + 3 * (size_t)hs[k], v3_normalized(v3_add(v3_load(vecmap + 3 * (size_t)hs[k]), dir)));
  }
  if (u3tmp < u3) {                                            /* :298 */
    distances[v3h] = u3tmp;                                    /* :300 */
    if ((u1 != 0 || u2 != 0) && vecmap) {                      /* :302 */
      const vec3 va = v3_load(vecmap + 3 * (size_t)v1h), vb = v3_load(vecmap + 3 * (size_t)v2h);
      v3_store(vecmap + 3 * (size_t)v3h, v3_normalized(v3_add(v3_scale(va, d31), v3_scale(vb, d32)))); /* :308 */
    }
    return u1 <= max_distance && u2 <= max_distance;           /* :311 */
  }
  return 0;
}

/* :341-491 */
void mo_inflation(const mo_mesh* m, const mo_inflation_cfg* cfg, const uint8_t* lethal,
                  const uint8_t* invalid, const float* edge_dist, float* cost_out, float* dist_out,
                  float* vec_out)
{
  const uint32_t V = m->V;
  uint8_t* fixed = (uint8_t*)calloc((size_t)V + 1, 1);         /* :386 */
  for (uint32_t v = 0; v < V; ++v) dist_out[v] = INFINITY;     /* :389 distances_.clear() */
  memset(vec_out, 0, sizeof(float) * 3 * (size_t)V);           /* :392 */
  mo_meap* pq = mo_meap_create(V);
  for (uint32_t v = 0; v < V; ++v)                             /* :397-402 (std::set order = ascending) */
    if (lethal[v]) { dist_out[v] = 0.0f; fixed[v] = 1; mo_meap_insert(pq, v, 0); }
  const float max_distance = (float)cfg->inflation_radius;     /* :438 passes config_.inflation_radius as const float& */
  while (!mo_meap_empty(pq)) {                                 /* :407 */
    const uint32_t cur = mo_meap_pop_min(pq, NULL);
    if (invalid && invalid[cur]) continue;                     /* :417 */
    fixed[cur] = 1;                                            /* :422 */
    /* :423 pmp_mesh.vertices(cur): CONVENTION neighbours in ascending edge id */
    for (uint32_t i = m->ve_ptr[cur]; i < m->ve_ptr[cur + 1]; ++i) {
      const uint32_t e = m->ve[i];
      const uint32_t nh = m->ev[2 * e] == cur ? m->ev[2 * e + 1] : m->ev[2 * e];
      /* :426-427 faces left of halfedge cur->nh, then of the opposite one.
       * CONVENTION: the face that lists (cur,nh) in winding order first. */
      uint32_t fs[2] = { m->ef[2 * e], m->ef[2 * e + 1] };
      if (fs[0] != NONE && fs[1] != NONE) {
        int first_has = 0;
        for (int k = 0; k < 3; ++k)
          if (m->fv[3 * fs[0] + k] == cur && m->fv[3 * fs[0] + (k + 1) % 3] == nh) first_has = 1;
        if (!first_has) { uint32_t t = fs[0]; fs[0] = fs[1]; fs[1] = t; }
      }
      for (int s = 0; s < 2; ++s) {
        const uint32_t fh = fs[s];
        if (fh == NONE) continue;                              /* :429-432 */
        /* :436-442 CONVENTION: (a,b,c) = stored face vertex order */
        const uint32_t a = m->fv[3 * fh], b = m->fv[3 * fh + 1], c = m->fv[3 * fh + 2];
        if (fixed[a] && fixed[b] && fixed[c]) continue;        /* :444 */
        else if (fixed[a] && fixed[b] && !fixed[c]) {          /* :448 */
          if (mo_inflation_wavefront_update(m, dist_out, vec_out, max_distance, edge_dist, a, b, c))
            mo_meap_insert(pq, c, dist_out[c]);
        } else if (fixed[a] && !fixed[b] && fixed[c]) {        /* :456 */
          if (mo_inflation_wavefront_update(m, dist_out, vec_out, max_distance, edge_dist, c, a, b))
            mo_meap_insert(pq, b, dist_out[b]);
        } else if (!fixed[a] && fixed[b] && fixed[c]) {        /* :464 */
          if (mo_inflation_wavefront_update(m, dist_out, vec_out, max_distance, edge_dist, b, c, a))
            mo_meap_insert(pq, a, dist_out[a]);
        }
      }
    }
  }
  mo_meap_destroy(pq); free(fixed);
  for (uint32_t v = 0; v < V; ++v)                             /* :484-490; default 0 (inflation_layer.h:74-77) */
    cost_out[v] = isinf(dist_out[v]) ? 0.0f : mo_inflation_fading(cfg, dist_out[v]);
}

/* combination_layer.cpp:44-85 (max) / :185-248 (weighted sum) */
void mo_combine(uint32_t V, int mode, int n_layers, const float* const* layers, const float* weights,
                float* out)
{
  for (uint32_t v = 0; v < V; ++v) {
    float cost = 0.0f;                                         /* defaultValue() combination_layer.h:52,94 */
    for (int l = 0; l < n_layers; ++l) {
      const float tmp = layers[l][v];
      if (mode == 0) cost = (cost < tmp) ? tmp : cost;         /* std::max :66 */
      else cost += weights[l] * tmp;                           /* :206 */
    }
    out[v] = cost;
  }
}
