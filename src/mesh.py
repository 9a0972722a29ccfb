from hidenn_fem_amd.mesh import generate_mesh, generate_mesh_gmsh, structured_tri_mesh  # noqa: F401
