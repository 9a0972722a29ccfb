"""Drop-in alias: the reference's module names (``src.models``, ``src.loss``, ``src.utils``,
``src.mesh``) re-exported from ``hidenn_fem_amd`` so scripts written for
achraf-15/HiDeNN-FEM import unchanged (option A)."""
