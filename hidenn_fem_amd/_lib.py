"""ctypes binding of ``libhidenn_hip.so`` (``include/hidenn_fem.h``).

There is no CPU fallback: if the shared library is missing, or a kernel is
asked to run on a tensor that is not on a ROCm device, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# HFEM_LAB=1 (set by the tools under scripts/): the lab build of the same library (ablations, stamps, staggers, variant
# kernels; `python hidenn_fem_amd/csrc/build.py --lab`).  The product never loads it.
LIB_PATH = os.path.join(_HERE, "csrc", "libhidenn_hip_lab.so" if os.environ.get("HFEM_LAB") == "1" else "libhidenn_hip.so")

_vp = C.c_void_p
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double


class PlanStats(C.Structure):
    _fields_ = [
        ("n_elems", _i64), ("n_nodes", _i64), ("n_edges", _i64),
        ("n_tiles", _i32), ("tile_elems", _i32),
        ("tile_elem_total", _i64), ("tile_node_total", _i64),
        ("max_tile_nodes", _i32), ("max_tile_owned", _i32),
        ("max_tile_elems", _i32), ("max_tile_edges", _i32),
        ("device_bytes", _i64), ("lds_bytes", _i32),
        ("shards", _i32), ("threads_per_tile", _i32), ("paired", _i32), ("slot_rows", _i32),
        ("store_policy", _i32), ("nodes_per_elem", _i32), ("row_line_factor", _f64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class AdamTensor(C.Structure):
    """``hfem_adam_tensor`` (include/hidenn_fem.h): one entry of the multi-tensor Adam table."""
    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("n", _i64), ("lr", _f64), ("beta1", _f64),
                ("beta2", _f64), ("eps", _f64), ("dtype", _i32), ("block_begin", _i32)]


# name -> (restype, argtypes); must list every symbol include/hidenn_fem.h declares
PROTOTYPES = {
    "hfem_version": (C.c_int, []),
    "hfem_last_error": (C.c_char_p, []),
    "hfem_device_count": (C.c_int, []),
    "hfem_tri3_energy_atomic": (C.c_int, [C.c_int, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _f64, _vp, _vp, _vp, _vp, _vp]),
    "hfem_edge2_energy_atomic": (C.c_int, [C.c_int, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hfem_plan_create": (C.c_int, [C.c_int, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i32, C.POINTER(_vp)]),
    "hfem_plan_create_ex": (C.c_int, [C.c_int, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i32, C.POINTER(_vp)]),
    "hfem_plan_destroy": (C.c_int, [_vp]),
    "hfem_plan_get_stats": (C.c_int, [_vp, C.POINTER(PlanStats)]),
    "hfem_plan_export": (_i64, [_vp, C.c_int, _vp, _i64]),
    "hfem_plan_serialize": (_i64, [_vp, _vp, _i64]),
    "hfem_plan_deserialize": (C.c_int, [C.c_int, _vp, _i64, C.POINTER(_vp)]),
    "hfem_tri3_energy_plan": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "hfem_tri3_energy_plan_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "hfem_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "hfem_get_option": (C.c_int, [C.c_char_p]),
    "hfem_tri3_eval_fwd": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "hfem_tri3_eval_bwd": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hfem_tri3_eval_fwd_conv": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp]),
    "hfem_tri3_eval_bwd_conv": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "hfem_edge2_eval_fwd": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "hfem_edge2_eval_bwd": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "hfem_quad4_energy_atomic": (C.c_int, [C.c_int, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "hfem_adam_step_dev": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _i64, _i32, C.c_double, C.c_double, C.c_double,
                                     C.c_double, _vp, _vp]),
    "hfem_counter_add": (C.c_int, [C.c_int, _vp, _i64, _vp]),
    "hfem_adam_multi_dev": (C.c_int, [C.c_int, _vp, _i32, _i32, _i64, _vp, _i64, _vp, _vp]),
    "hfem_lbfgs_create": (C.c_int, [C.c_int, _i64, _i32, _i32, C.POINTER(_vp)]),
    "hfem_lbfgs_destroy": (C.c_int, [_vp]),
    "hfem_lbfgs_check": (C.c_int, [_vp, _vp, _vp, _i32, C.c_double, C.c_double, _vp, _vp]),
    "hfem_lbfgs_direction": (C.c_int, [_vp, _vp, C.c_double, C.c_double, _vp]),
    "hfem_lbfgs_apply": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "hfem_lbfgs_direction_ptr": (_vp, [_vp]),
    "hfem_lbfgs_shard_payload_doubles": (_i64, [_vp]),
    "hfem_lbfgs_shard_gather": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "hfem_lbfgs_shard_local": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "hfem_lbfgs_shard_finish": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _f64, _f64, _f64, _vp, _vp]),
    "hfem_lbfgs_shard_apply": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "hfem_lbfgs_shard_status": (C.c_int, [_vp, _vp, _vp]),
    "hfem_tri3_von_mises": (C.c_int, [C.c_int, _vp, _vp, _vp, _i64, C.c_double, C.c_double, _vp, _vp, _vp]),
    "hfem_line2_slopes": (C.c_int, [C.c_int, _vp, _vp, _i64, _i32, _vp, _vp]),
    "hfem_tri3_energy_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                             _f64, _f64, _f64, _f64, _f64, _vp, _vp, _i32, _vp]),
    "hfem_tri3_energy_adam_step_ex": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                _vp, _f64, _f64, _f64, _f64, _f64, _vp, _i32, _i32, _vp, _i32, _vp]),
    "hfem_adam_prep": (C.c_int, [C.c_int, _vp, _f64, _f64, _vp, _vp]),
    "hfem_plan_loss_sum": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "hfem_iface_pack": (C.c_int, [C.c_int, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "hfem_iface_unpack": (C.c_int, [C.c_int, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i64, _i64, _vp, _vp]),
    "hfem_mg_load": (C.c_int, [C.c_char_p]),
    "hfem_mg_unique_id": (C.c_int, [_vp]),
    "hfem_mg_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "hfem_mg_comm_destroy": (C.c_int, [_vp]),
    "hfem_mg_allreduce_sum": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "hfem_mg_allgather": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "hfem_adam_step_rows_dev": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64, _vp, _vp]),
    "hfem_plan_set_span_stamps": (C.c_int, [_vp, _vp, _i64]),
    "hfem_peer_create": (C.c_int, [C.c_int, _i32, _i32, _i64, C.POINTER(_vp)]),
    "hfem_peer_ipc_handle": (C.c_int, [_vp, _vp]),
    "hfem_peer_connect": (C.c_int, [_vp, _vp]),
    "hfem_peer_destroy": (C.c_int, [_vp]),
    "hfem_peer_attach_get": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i64, _vp, _i64]),
    "hfem_plan_set_peer_get": (C.c_int, [_vp, _vp, _i32, _i32]),
    "hfem_peer_attach_put": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _f64, _f64]),
    "hfem_plan_set_peer_put": (C.c_int, [_vp, _vp, _vp, _vp]),
    "hfem_peer_status": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "hfem_plan_iface_pack_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _f64, _f64, _vp, _vp]),
    "hfem_iface_unpack_f32": (C.c_int, [C.c_int, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i64, _i64, _vp, _vp]),
    "hfem_plan_iface_put_f32": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i64, _vp, _f64, _f64, _vp, _vp]),
    "hfem_peer_iface_get_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp]),
    "hfem_plan_iface_put": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i64, _vp, _f64, _f64, _vp, _vp]),
    "hfem_peer_iface_get": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp]),
    "hfem_plan_iface_pack": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _f64, _f64, _vp, _vp]),
    "hfem_adam_step_rows2_dev": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp, _vp, _i64, _f64,
                                           _f64, _f64, _f64, _vp, _i64, _vp]),
    "hfem_adam_step_rows2_dev_f32": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp, _vp, _i64, _f64,
                                               _f64, _f64, _f64, _vp, _i64, _vp]),
    "hfem_quad4_energy_plan": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "hfem_quad4_energy_plan_body": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "hfem_quad4_energy_plan_ex": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "hfem_quad4_eval_fwd": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "hfem_quad4_eval_bwd": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hfem_adam_step": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _i64, _i32, _f64, _f64, _f64, _f64, _i64, _vp]),
    "hfem_scatter_rows": (C.c_int, [C.c_int, _vp, _vp, _i64, _i32, _vp, _vp]),
    "hfem_gather_rows": (C.c_int, [C.c_int, _vp, _vp, _i64, _i32, _vp, _vp]),
    "hfem_grid_param_fwd": (C.c_int, [C.c_int, _vp, _i64, _f64, _f64, _vp, _vp, _vp, _vp]),
    "hfem_grid_param_bwd": (C.c_int, [C.c_int, _vp, _i64, _f64, _f64, _vp, _vp, _vp, _vp]),
    "hfem_grid_param_ws_elems": (_i64, [_i64]),
    "hfem_grid_param_fwd_ws": (C.c_int, [C.c_int, _vp, _i64, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hfem_grid_param_bwd_ws": (C.c_int, [C.c_int, _vp, _i64, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hfem_line2_eval_fwd": (C.c_int, [C.c_int, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "hfem_line2_eval_bwd": (C.c_int, [C.c_int, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hfem_bar_energy": (C.c_int, [C.c_int, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp]),
    "hfem_line2_mse": (C.c_int, [C.c_int, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "hfem_rectq4_eval_fwd": (C.c_int, [C.c_int, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "hfem_rectq4_eval_bwd": (C.c_int, [C.c_int, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hfem_rectq4_mse": (C.c_int, [C.c_int, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
}

# float-row twins of the 1D / structured entry points (same argument lists; every array pointer is float* except the
# fp64 scratch of the *_ws forms)
_F32_TWINS = ["hfem_grid_param_fwd", "hfem_grid_param_bwd", "hfem_grid_param_fwd_ws", "hfem_grid_param_bwd_ws",
              "hfem_line2_eval_fwd", "hfem_line2_eval_bwd", "hfem_bar_energy", "hfem_line2_mse", "hfem_rectq4_eval_fwd",
              "hfem_rectq4_eval_bwd", "hfem_rectq4_mse"]
PROTOTYPES.update({name + "_f32": PROTOTYPES[name] for name in _F32_TWINS})

_lib = None


def lib():
    """Load the HIP extension (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"hidenn_fem_amd: HIP extension not built ({LIB_PATH} missing). "
                "Run `python hidenn_fem_amd/csrc/build.py` (or __graft_entry__.build()). "
                "There is no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().hfem_last_error()
        raise RuntimeError(f"libhidenn_hip {what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def require_gpu_tensor(t: torch.Tensor, name: str, dtype=torch.float64):
    """The kernels only run on ROCm device memory; anything else is an error, not a fallback."""
    if not t.is_cuda:
        raise RuntimeError(
            f"hidenn_fem_amd: `{name}` is on {t.device}; the HIP kernels need a ROCm (cuda) tensor. "
            "There is no CPU fallback -- move the model/inputs to the GPU.")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"hidenn_fem_amd: `{name}` must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"hidenn_fem_amd: `{name}` must be contiguous")
    return t


def ptr(t):
    """Raw device/host pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr(device: torch.device):
    return torch.cuda.current_stream(device).cuda_stream


def dev_index(device: torch.device) -> int:
    return device.index if device.index is not None else torch.cuda.current_device()
