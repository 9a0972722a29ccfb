"""hidenn_fem_amd -- MI355X-native HiDeNN-FEM element-evaluation / energy engine.

Host-side mirror of the reference's construction API (``src/models.py``,
``src/loss.py``, ``src/utils.py``, ``src/mesh.py``) over hand-written gfx950 HIP
kernels reached through a C ABI (``include/hidenn_fem.h``).  No CPU fallback:
the compute paths raise if the HIP extension or a ROCm device is missing.
"""
__version__ = "0.1.0"
