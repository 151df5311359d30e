import numpy as np

from mesh_navigation_amd import meshgen
from oracle import oracle as O


def test_sizes_of_survey_table():
    m = meshgen.terrain(224, 0.1, 1)           # config "50k"
    assert (m.V, m.F, m.E) == (50176, 99458, 149633)


def test_edge_convention_matches_oracle():
    for mesh in (meshgen.terrain(17, 0.1, 4), meshgen.flat_grid(6)):
        om = O.OracleMesh(mesh.xyz, mesh.faces)
        assert np.array_equal(om.edges(), mesh.edges)
        assert np.array_equal(om.face_edges(), mesh.face_edges)
        assert np.array_equal(om.edge_distances().view(np.uint32), meshgen.edge_lengths(mesh).view(np.uint32))


def test_terrain_is_seeded_and_ccw():
    a, b = meshgen.terrain(20, 0.1, 9), meshgen.terrain(20, 0.1, 9)
    assert np.array_equal(a.xyz, b.xyz)
    assert not np.array_equal(a.xyz, meshgen.terrain(20, 0.1, 10).xyz)
    om = O.OracleMesh(a.xyz, a.faces)
    assert (om.face_normals()[:, 2] > 0).all()


def test_grid_edges_closed_form_matches_first_appearance_order():
    """meshgen.grid_edges (no sort; the 10M-vertex bench mesh) == edges_from_faces(grid_faces) (lvr2/pmp edge ids)"""
    from mesh_navigation_amd import meshgen
    for N in (2, 3, 7, 33):
        e1, fe1 = meshgen.edges_from_faces(meshgen.grid_faces(N))
        e2, fe2 = meshgen.grid_edges(N)
        assert np.array_equal(e1, e2) and np.array_equal(fe1, fe2)
