/* mnav_planner_host.hpp -- the host half of MeshPlanner::makePlan, written once.
 *
 * libmnav has two front ends that present the reference's planner classes: the ROS 2 plugin package
 * (integration/mesh_gpu_planners: lvr2 handles, the reference's own MeshMap) and its ROS-free twin
 * (mesh_navigation_amd/csrc/adapter: plain ids, a host mesh with the same queries).  Everything the two do between the
 * C ABI and the `std::vector<PoseStamped>` they return follows the same lines of the reference, so it lives here as
 * templates over the front end's vector / pose / handle types and is instantiated by both:
 *
 *   vertex_path_poses   dijkstra_mesh_planner.cpp:89-116   one pose per path vertex, looking at the next one
 *   face_path_poses     cvp_mesh_planner.cpp:99-124        one pose per back-tracking step, the goal pose verbatim at the end
 *   backtrack_on_host   cvp_mesh_planner.cpp:920-966       the walk along the vector field with the map's own meshAhead
 *   dijkstra_vertex_path   dijkstra_mesh_planner.cpp:287-373 on the device: mnav_plan_dijkstra, vertex ids back
 *   cvp_field_is_set    cvp_mesh_planner.cpp:722-724, :238  which entries of the device's vector map the map has to hold
 *
 * Header-only; needs nothing but the C ABI (mnav.h) and the standard library.  Result codes are mbf_msgs GetPath::Result
 * values as mnav.h defines them. */
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "mnav.h"

namespace mnav_host {

/* dijkstra_mesh_planner.cpp:89-116.  `path`: vertex handles, robot side first.  `here`: the robot position; `target`: the
 * goal position (the last pose looks at it).  pose_from(from, to, normal, length&) is mesh_map::calculatePoseFromPosition. */
template <class PoseStamped, class Vector, class Path, class PositionOf, class NormalOf, class PoseFrom>
void vertex_path_poses(const Path& path, Vector here, const Vector& target, const PoseStamped& stamped, PositionOf position_of,
                       NormalOf normal_of, PoseFrom pose_from, std::vector<PoseStamped>& plan, double& cost)
{
  cost = 0;                                                           // :89
  if (path.empty()) return;                                           // :90
  auto up = normal_of(path.front());                                  // :94
  float step = 0.f;
  PoseStamped pose = stamped;
  for (const auto vH : path) {                                        // :100-111
    const Vector next = position_of(vH);
    pose.pose = pose_from(here, next, up, step);
    cost += step;
    here = next;
    up = normal_of(vH);
    plan.push_back(pose);
  }
  pose.pose = pose_from(here, target, up, step);                      // :113
  cost += step;
  plan.push_back(pose);
}

/* cvp_mesh_planner.cpp:99-124.  `path`: (position, face) pairs, robot side first; skipped altogether after a cancel (:101). */
template <class PoseStamped, class Path, class GoalPose, class FaceNormalOf, class PoseFrom>
void face_path_poses(const Path& path, bool cancelled, const GoalPose& goal_pose, const PoseStamped& stamped,
                     FaceNormalOf face_normal_of, PoseFrom pose_from, std::vector<PoseStamped>& plan, double& cost)
{
  cost = 0;                                                           // :99
  if (cancelled || path.empty()) return;                              // :101
  auto it = path.begin();
  auto here = it->first;                                              // :103
  auto face = it->second;                                             // :104
  float step = 0.f;
  PoseStamped pose = stamped;
  for (++it; it != path.end(); ++it) {                                // :108-117
    pose.pose = pose_from(here, it->first, face_normal_of(face), step);
    cost += step;
    here = it->first;
    face = it->second;
    plan.push_back(pose);
  }
  pose.pose = goal_pose;                                              // :119-123: the goal pose itself closes the plan
  plan.push_back(pose);
}

/* cvp_mesh_planner.cpp:920-966.  `mesh_ahead(pos&, face&, step_width)`: 1 = advanced, 0 = no way on (:937-941), -1 = the
 * half-edge mesh panicked (:944-949; the caller catches its exception type).  `path` gets (position, face) pairs pushed at
 * the front: seed side first when done.  `max_steps` = 0: no limit (the reference has none). */
template <class Vector, class Face, class Path, class Cancelled, class MeshAhead>
uint32_t backtrack_on_host(const Vector& seed, Face seed_face, const Vector& target, Face target_face, double step_width,
                           Cancelled cancelled, MeshAhead mesh_ahead, size_t max_steps, Path& path, std::string& message)
{
  Face face = target_face;                                            // :920-922
  Vector pos = target;
  path.push_front(std::make_pair(pos, face));                         // :924
  size_t steps = 0;
  while (pos.distance2(seed) > step_width && !cancelled()) {          // :927 (squared distance against the width, as is)
    const int ahead = mesh_ahead(pos, face, step_width);              // :933
    if (ahead < 0) { message = "Could not find a valid path, while back-tracking from the goal: HalfEdgeMesh panicked!"; return MNAV_NO_PATH_FOUND; }
    if (ahead == 0) { message = "Could not find a valid path, while back-tracking from the goal"; return MNAV_NO_PATH_FOUND; }   // :939
    path.push_front(std::make_pair(pos, face));                       // :935
    if (max_steps && ++steps > max_steps) { message = "vector field back-tracking does not terminate"; return MNAV_NO_PATH_FOUND; }
  }
  path.push_front(std::make_pair(seed, seed_face));                   // :951
  if (cancelled()) return MNAV_CANCELED;                              // :962-966
  return MNAV_SUCCESS;
}

/* dijkstra_mesh_planner.cpp:287-373 on the device.  Potential, predecessors and the vector map stay in HBM (fetched with
 * mnav_download_output when somebody reads them); `ids`: the vertex path, seed side first (:358-373). */
inline uint32_t dijkstra_vertex_path(mnav_ctx* ctx, uint32_t seed_vertex, uint32_t target_vertex, double goal_dist_offset, double cost_limit,
                                     uint32_t V, std::vector<uint32_t>& ids)
{
  ids.assign(V ? V : 1, 0u);
  uint32_t n = 0;
  const uint32_t code = mnav_plan_dijkstra(ctx, seed_vertex, target_vertex, goal_dist_offset, cost_limit, nullptr, nullptr, ids.data(), V, &n, nullptr);
  ids.resize(code == MNAV_SUCCESS ? n : 0);
  return code;
}

/* MeshMap::setVectorMap after a CVP plan (:238): the field holds the three seed vertices (their offset from the seed
 * position, :722-724, whatever its value) and every vertex the wave updated; the device writes all-zero entries elsewhere. */
inline bool cvp_field_is_set(const float* vector_map, uint32_t v, const uint32_t seed_face_vertices[3])
{
  const float* q = vector_map + 3 * (size_t)v;
  return q[0] != 0.f || q[1] != 0.f || q[2] != 0.f || v == seed_face_vertices[0] || v == seed_face_vertices[1] || v == seed_face_vertices[2];
}

}  // namespace mnav_host
