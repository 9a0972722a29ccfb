from hidenn_fem_amd.post import compute_du_dx_per_element, von_mises  # noqa: F401  (compute cores of src/plots.py)
