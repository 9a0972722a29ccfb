"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the HiDeNN-FEM hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / reported CPU baseline.  The product
package (``hidenn_fem_amd``) never imports from here and has no CPU fallback.

Contents
--------
``ref_chain``    op-for-op PyTorch (CPU, autograd) restatement of the reference
                 op chain (``/root/reference/src/{models,loss,utils}.py`` and the
                 inline losses of ``examples/example{1,2,3}.py``); each function
                 cites the file:line it follows.
``hfem_oracle.c`` plain-C closed forms (forward + hand backward) of the same
                 path, compiled by ``oracle/build.py`` to ``libhfem_oracle.so``.
``closed_form``  ctypes front-end for the C restatement.

Parity pin: both restatements are checked against golden vectors produced by
importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.
"""
