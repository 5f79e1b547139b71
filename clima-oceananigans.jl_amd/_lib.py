"""ctypes binding of libocnhip.so (the C ABI declared in include/ocnhip.h).

The library is the product: if it is missing this module raises -- there is no CPU fallback.
``OCNHIP_LIB`` may point at another build of the *same sources* (tests/hostemu sets it to the
host-emulation build so that kernels can be exercised on a machine without a GPU; see
csrc/compat.h).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_TRACERS = 8

# enums (include/ocnhip.h)
PERIODIC, BOUNDED, FLAT = 0, 1, 2
CENTER, FACE = 0, 1
ADV_NONE, ADV_C2, ADV_C4, ADV_U5, ADV_WENO5_Z, ADV_WENO5_JS, ADV_U1, ADV_U3 = range(8)
STEPPER_AB2, STEPPER_RK3 = 0, 1
CLOSURE_NONE, CLOSURE_SCALAR, CLOSURE_AMD = 0, 1, 2
BUOYANCY_NONE, BUOYANCY_TRACER, BUOYANCY_LINEAR_TS = 0, 1, 2
BC_DEFAULT, BC_PERIODIC, BC_NOFLUX, BC_FLUX, BC_VALUE, BC_GRADIENT, BC_IMPENETRABLE, BC_NONE = range(8)
WEST, EAST, SOUTH, NORTH, BOTTOM, TOP = range(6)
F_U, F_V, F_W, F_PHY, F_PNHS, F_GN, F_GM, F_TRACER, F_NU, F_KAPPA = 0, 1, 2, 3, 4, 16, 32, 48, 64, 72

ERRORS = {-1: "OCN_EINVAL", -2: "OCN_ENOMEM", -3: "OCN_EHIP", -4: "OCN_EUNSUPPORTED", -5: "OCN_ESTATE"}


class GridDesc(C.Structure):
    _fields_ = [("N", C.c_int32 * 3), ("H", C.c_int32 * 3), ("topology", C.c_int32 * 3),
                ("x0", C.c_double * 3), ("L", C.c_double * 3), ("z_faces", C.POINTER(C.c_double)),
                ("rank", C.c_int32), ("nranks", C.c_int32)]


class BC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("value", C.c_double), ("array", C.POINTER(C.c_double))]


class ModelDesc(C.Structure):
    _fields_ = [("advection", C.c_int32), ("stepper", C.c_int32), ("chi", C.c_double),
                ("n_tracers", C.c_int32), ("closure", C.c_int32), ("nu", C.c_double),
                ("kappa", C.c_double * MAX_TRACERS), ("amd_Cnu", C.c_double),
                ("amd_Ckappa", C.c_double * MAX_TRACERS), ("amd_Cb", C.c_double), ("amd_has_Cb", C.c_int32),
                ("coriolis_fplane", C.c_int32), ("f", C.c_double),
                ("buoyancy", C.c_int32), ("b_index", C.c_int32), ("T_index", C.c_int32), ("S_index", C.c_int32),
                ("g", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("bcs", (BC * 6) * (3 + MAX_TRACERS)), ("nu_bcs", BC * 6), ("kappa_bcs", (BC * 6) * MAX_TRACERS)]


class HydroDesc(C.Structure):
    _fields_ = [("free_surface", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("w", C.c_void_p), ("pHY", C.c_void_p),
                ("ntracers", C.c_int32), ("tracers", C.POINTER(C.c_void_p)), ("Gn", C.POINTER(C.c_void_p)), ("Gm", C.POINTER(C.c_void_p)),
                ("buoyancy_kind", C.c_int32), ("T_index", C.c_int32), ("S_index", C.c_int32),
                ("gravitational_acceleration", C.c_double), ("thermal_expansion", C.c_double), ("haline_contraction", C.c_double)]


class HGridDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("N", C.c_int32 * 3), ("H", C.c_int32 * 3), ("topology", C.c_int32 * 3),
                ("x0", C.c_double * 3), ("L", C.c_double * 3), ("z_faces", C.POINTER(C.c_double)), ("radius", C.c_double), ("partition", C.c_int32), ("band_overlap", C.c_int32)]


NOTHING = 2
HGRID_RECTILINEAR, HGRID_LATLON = 0, 1


class OcnError(RuntimeError):
    pass


_lib = None


ABI_VERSION = 5   # OCN_ABI_VERSION of include/ocnhip.h


def lib_path():
    return os.environ.get("OCNHIP_LIB") or os.path.join(HERE, "libocnhip.so")


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise OcnError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(path)
    P, I, D = C.c_void_p, C.c_int, C.c_double
    PD = C.POINTER(C.c_double)
    sig = {
        "ocn_abi_version": (I, []),
        "ocn_init": (I, [I, C.POINTER(P)]),
        "ocn_destroy": (None, [P]),
        "ocn_sync": (I, [P]),
        "ocn_last_error": (C.c_char_p, [P]),
        "ocn_stream": (P, [P]),
        "ocn_grid_create": (I, [P, C.POINTER(GridDesc), C.POINTER(P)]),
        "ocn_grid_destroy": (None, [P]),
        "ocn_model_create": (I, [P, C.POINTER(ModelDesc), C.POINTER(P)]),
        "ocn_model_destroy": (None, [P]),
        "ocn_model_halo": (I, [P, C.POINTER(C.c_int32 * 3)]),
        "ocn_model_path": (I, [P, C.c_char_p, C.c_size_t]),
        "ocn_model_graph_replays": (I, [P, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
        "ocn_field_create": (I, [P, I, I, I, C.POINTER(P)]),
        "ocn_field_destroy": (None, [P]),
        "ocn_field_parent_shape": (I, [P, C.POINTER(C.c_int32 * 3), C.POINTER(C.c_int32 * 3), C.POINTER(C.c_int32 * 3)]),
        "ocn_field_parent_layout": (I, [P, C.POINTER(C.c_int64 * 3), C.POINTER(C.c_int64)]),
        "ocn_field_parent_ptr": (P, [P]),
        "ocn_field_parent_upload": (I, [P, PD]),
        "ocn_field_parent_download": (I, [P, PD]),
        "ocn_field_shape": (I, [P, I, C.POINTER(C.c_int32 * 3), C.POINTER(C.c_int32 * 3), C.POINTER(C.c_int32 * 3)]),
        "ocn_field_device_ptr": (P, [P, I]),
        "ocn_field_layout": (I, [P, I, C.POINTER(C.c_int64 * 3), C.POINTER(C.c_int64)]),
        "ocn_field_upload": (I, [P, I, PD]),
        "ocn_field_download": (I, [P, I, PD]),
        "ocn_field_set_interior": (I, [P, I, PD]),
        "ocn_field_get_interior": (I, [P, I, PD]),
        "ocn_fill_halos": (I, [P, C.c_uint32]),
        "ocn_update_state": (I, [P]),
        "ocn_compute_tendencies": (I, [P]),
        "ocn_ab2_step": (I, [P, D, D]),
        "ocn_rk3_substep": (I, [P, D, D, D, I]),
        "ocn_store_tendencies": (I, [P]),
        "ocn_pressure_correction": (I, [P, D]),
        "ocn_poisson_solve_host": (I, [P, PD, PD]),
        "ocn_pressure_correct_velocities": (I, [P, D]),
        "ocn_set_epilogue": (I, [P, I]),
        "ocn_time_step": (I, [P, D, I]),
        "ocn_clock": (I, [P, PD, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
        "ocn_set_clock": (I, [P, D, C.c_int64, D]),
        "ocn_max_abs_divergence": (I, [P, PD]),
        "ocn_comm_unique_id": (I, [P]),
        "ocn_comm_init": (I, [P, I, I, P]),
        "ocn_comm_probe": (I, [P, I, I, P, D]),
        "ocn_comm_rank": (I, [P, C.POINTER(I), C.POINTER(I)]),
        "ocn_hgrid_create": (I, [P, C.POINTER(HGridDesc), C.POINTER(P)]),
        "ocn_hgrid_destroy": (None, [P]),
        "ocn_hgrid_band": (I, [P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "ocn_hgrid_metric": (I, [P, I, PD, I]),
        "ocn_hfield_create": (I, [P, I, I, I, C.POINTER(P)]),
        "ocn_hfield_destroy": (None, [P]),
        "ocn_hfield_shape": (I, [P, C.POINTER(C.c_int32 * 3), C.POINTER(C.c_int32 * 3), C.POINTER(C.c_int32 * 3)]),
        "ocn_hfield_ptr": (P, [P]),
        "ocn_hfield_upload": (I, [P, PD]),
        "ocn_hfield_download": (I, [P, PD]),
        "ocn_hfield_fill_halos": (I, [P]),
        "ocn_sefs_create": (I, [P, D, I, C.POINTER(P)]),
        "ocn_sefs_destroy": (None, [P]),
        "ocn_sefs_field": (P, [P, I]),
        "ocn_sefs_set_weights": (I, [P, I, PD, PD]),
        "ocn_sefs_substep": (I, [P, D, I]),
        "ocn_sefs_substeps": (I, [P, D, I, I, I]),
        "ocn_sefs_graph_replays": (I, [P, C.POINTER(C.c_int64)]),
        "ocn_sefs_barotropic_mode": (I, [P, P, P, I]),
        "ocn_sefs_set_average_to_zero": (I, [P]),
        "ocn_sefs_corrector": (I, [P, P, P]),
        "ocn_sefs_step": (I, [P, P, P, P, P, D, D]),
        "ocn_hfield_ab2_step": (I, [P, P, P, D, D]),
        "ocn_hfield_store_tendency": (I, [P, P]),
        "ocn_hydro_compute_w": (I, [P, P, P]),
        "ocn_hydro_pressure": (I, [P, I, D, D, D, P, P]),
        "ocn_hydro_create": (I, [C.POINTER(HydroDesc), C.POINTER(P)]),
        "ocn_hydro_destroy": (None, [P]),
        "ocn_hydro_update_state": (I, [P]),
        "ocn_hydro_ab2_step": (I, [P, D, D]),
        "ocn_hydro_step_after_tendencies": (I, [P, D, D, I]),
        "ocn_hydro_set_physics": (I, [P, I, I, D, I]),
        "ocn_hydro_set_closure": (I, [P, D, I, PD]),
        "ocn_hydro_calculate_tendencies": (I, [P]),
        "ocn_hydro_time_step": (I, [P, D, I]),
        "ocn_profile_enable": (I, [P, I]),
        "ocn_profile_read": (I, [P, C.c_char_p, PD, C.POINTER(C.c_int64)]),
        "ocn_profile_reset": (I, [P]),
        "ocn_profile_filter": (I, [P, C.c_char_p]),
        "ocn_measure_copy_rate": (I, [P, C.c_size_t, I, PD]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError here = the library does not export what the header declares
        fn.restype, fn.argtypes = res, args
    if L.ocn_abi_version() != ABI_VERSION:   # the structs above are laid out for exactly this version of include/ocnhip.h
        raise OcnError(f"{path} reports ABI version {L.ocn_abi_version()}, this binding is written for {ABI_VERSION}: rebuild")
    _lib = L
    _lib._signatures = sig
    return _lib


def check(rc, ctx=None):
    if rc != 0:
        msg = load().ocn_last_error(ctx).decode(errors="replace")
        raise OcnError(f"{ERRORS.get(rc, rc)}: {msg}")
