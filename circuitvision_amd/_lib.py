"""ctypes binding of libcvmi355.so (include/cvmi355.h).  There is NO fallback: if the HIP library is
missing or a call fails, an ordinary Python exception is raised (the reference's callers catch
broadly: /root/reference/src/circuit_analyzer.py:255-263, :381-386)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CVMI_LIB_PATH: A/B measurements load an alternative build of the same library (tools/ab_env.sh) instead of overwriting the shipped file
LIB_PATH = os.environ.get("CVMI_LIB_PATH") or os.path.join(_HERE, "libcvmi355.so")

F16, F32, BF16 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_RELU, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3, 4


class CvmiError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p), ("x1", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("res", C.c_void_p), ("y", C.c_void_p),
        ("x0_ld", C.c_int), ("x1_ld", C.c_int), ("res_ld", C.c_int), ("y_ld", C.c_int),
        ("c0", C.c_int), ("c1", C.c_int), ("up0", C.c_int), ("up1", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("OH", C.c_int), ("OW", C.c_int),
        ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("N", C.c_int), ("Kpad", C.c_int), ("act", C.c_int), ("dtype", C.c_int),
        ("out_f32", C.c_int), ("scalar_gather", C.c_int),
        ("res_mod", C.c_int), ("act_after_res", C.c_int), ("shuffle_cout", C.c_int), ("res_rep", C.c_int),
        ("row_stats", C.c_void_p),
    ]


class C3k2Desc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p),
        ("w0", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p), ("w3", C.c_void_p),
        ("b0", C.c_void_p), ("b1", C.c_void_p), ("b2", C.c_void_p), ("b3", C.c_void_p),
        ("x_ld", C.c_int), ("y_ld", C.c_int), ("kpad0", C.c_int), ("kpad1", C.c_int), ("kpad2", C.c_int), ("kpad3", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("c1", C.c_int), ("c", C.c_int), ("h", C.c_int), ("c2", C.c_int),
        ("fuse_cv1", C.c_int), ("shortcut", C.c_int), ("dtype", C.c_int),
    ]


class DwPwDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("wd", C.c_void_p), ("bd", C.c_void_p),
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("x_ld", C.c_int), ("y_ld", C.c_int), ("kpad1", C.c_int), ("kpad2", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("N1", C.c_int), ("N2", C.c_int),
        ("dtype", C.c_int),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("q_sb", C.c_longlong), ("q_sh", C.c_longlong), ("q_st", C.c_longlong),
        ("k_sb", C.c_longlong), ("k_sh", C.c_longlong), ("k_st", C.c_longlong),
        ("v_sb", C.c_longlong), ("v_sh", C.c_longlong), ("v_st", C.c_longlong),
        ("o_sb", C.c_longlong), ("o_sh", C.c_longlong), ("o_st", C.c_longlong),
        ("B", C.c_int), ("heads", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int),
        ("dqk", C.c_int), ("dv", C.c_int), ("scale", C.c_float), ("dtype", C.c_int),
        ("win", C.c_int), ("grid_h", C.c_int), ("grid_w", C.c_int), ("q_pool", C.c_int),
        ("q_bdiv", C.c_int), ("kv_bdiv", C.c_int), ("av_fp8", C.c_int), ("q_log2", C.c_int),
    ]


# name -> (restype, argtypes); every symbol declared in include/cvmi355.h
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    "cvmi_version": (_i, []),
    "cvmi_last_error": (C.c_char_p, []),
    "cvmi_last_kernel": (C.c_char_p, []),
    "cvmi_device_info": (_i, [_i, C.POINTER(_i)]),
    "cvmi_desc_size": (C.c_size_t, [_i]),
    "cvmi_graph_begin": (_i, [_vp]),
    "cvmi_graph_end": (_i, [_vp, C.POINTER(_vp)]),
    "cvmi_graph_launch": (_i, [_vp, _vp]),
    "cvmi_graph_destroy": (_i, [_vp]),
    "cvmi_conv2d": (_i, [C.POINTER(ConvDesc), _vp]),
    "cvmi_c3k2_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "cvmi_c3k2": (_i, [C.POINTER(C3k2Desc), _vp]),
    "cvmi_stem2_supported": (_i, [_i, _i, _i]),
    "cvmi_stem2": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cvmi_dwpw_supported": (_i, [_i, _i, _i, _i]),
    "cvmi_dwpw": (_i, [C.POINTER(DwPwDesc), _vp]),
    "cvmi_dwconv3x3": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cvmi_sppf_pool": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cvmi_attention": (_i, [C.POINTER(AttnDesc), _vp]),
    "cvmi_detect_decode": (_i, [C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i),
                                C.POINTER(_i), C.POINTER(_f), _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "cvmi_yolo_nms_best": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "cvmi_yolo_nms_workspace": (C.c_size_t, [_i, _i]),
    "cvmi_yolo_nms": (_i, [_vp, _i, _i, _i, _f, _f, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "cvmi_letterbox": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cvmi_letterbox_batch": (_i, [_vp, _i, _i, _i, _vp, C.c_longlong, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cvmi_nchw_to_nhwc": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cvmi_nhwc_to_nchw_f32": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "cvmi_layernorm": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, C.c_longlong, _i, _f, _i, _i, _i, _i, _i, _vp]),
    "cvmi_layernorm_dual": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, C.c_longlong, _i, _f, _vp]),
    "cvmi_hiera_mlp_supported": (_i, [_i]),
    "cvmi_hiera_mlp_packed_bytes": (C.c_size_t, [_i]),
    "cvmi_hiera_mlp": (_i, [_vp, _i, _vp, _vp, _f, _vp, _vp, C.c_longlong, _i, _i, _vp]),
    "cvmi_hiera_mlp_stats": (_i, [_vp, _i, _vp, _vp, _f, _vp, _vp, C.c_longlong, _i, _i, _vp, _f, _vp]),
    "cvmi_tok_linear_supported": (_i, [_i]),
    "cvmi_tok_linear_packed_bytes": (C.c_size_t, [_i, _i]),
    "cvmi_tok_linear_format": (_i, [_i]),
    "cvmi_tok_linear_stats_parts": (_i, [C.c_longlong, _i, _i]),
    "cvmi_tok_linear": (_i, [_vp, _i, _i, _vp, _vp, _f, _vp, _vp, _i, _i, C.c_longlong, _i, _i, _i, _i, _vp]),
    "cvmi_tok_linear_stats": (_i, [_vp, _i, _i, _vp, _vp, _f, _vp, _vp, _i, _i, C.c_longlong, _i, _i, _i, _i, _vp, _i, _vp, _f, _vp]),
    "cvmi_tok_linear_pool_stats": (_i, [_vp, _i, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "cvmi_tok_linear_pool": (_i, [_vp, _i, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cvmi_debug_stamps": (_i, [C.POINTER(C.c_ulonglong)]),
    "cvmi_maxpool2x2": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cvmi_space_to_depth4": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "cvmi_cast": (_i, [_vp, _i, _i, _vp, _i, _i, C.c_longlong, _i, _vp]),
    "cvmi_prompt_tokens": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp]),
    "cvmi_repeat_images": (_i, [_vp, _vp, C.c_longlong, _i, _i, _vp]),
    "cvmi_hyper_masks": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _f, _vp]),
    "cvmi_select_mask": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _i, _i, _vp]),
    "cvmi_bilinear_f32": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _f, _vp]),
    "cvmi_mask_extent": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "cvmi_mask_postprocess": (_i, [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "cvmi_mask_postprocess_sizes": (_i, [_vp, _i, _i, _i, _vp, _f, _vp, _vp, _vp]),
    "cvmi_upsample_refine": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, C.POINTER(_i), _i, _i, _vp]),
    "cvmi_sam2_transform": (_i, [_vp, _i, _i, _vp, _i, _i, _vp]),
    "cvmi_sam2_transform_batch": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "cvmi_sam2_transform_rects": (_i, [_vp, C.c_longlong, _i, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
}

_lib = None


def load():
    """Load the shared library once; raise CvmiError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CvmiError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C circuitvision_amd/csrc`).  There is no CPU fallback.")
    # torch first: it carries its own HIP runtime (libamdhip64) and the process must hold exactly one;
    # loading ours first would bring in /opt/rocm's copy beside torch's ("no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    check_abi(lib)
    _lib = lib
    return lib


# kind (include/cvmi355.h: CVMI_DESC_*) -> the ctypes mirror of that descriptor struct
DESC_MIRRORS = {0: ConvDesc, 1: C3k2Desc, 2: DwPwDesc, 3: AttnDesc}


def check_abi(lib, mirrors=None):
    """Every descriptor mirror must have exactly the size the library was compiled with: a mirror that is short by a trailing
    field would make the kernels' host code read past its end (cvmi_desc_size, include/cvmi355.h)."""
    for kind, cls in (mirrors or DESC_MIRRORS).items():
        want = int(lib.cvmi_desc_size(kind))
        if want != C.sizeof(cls):
            raise CvmiError(f"ABI mismatch: {cls.__name__} is {C.sizeof(cls)} bytes in this binding but {want} bytes in {LIB_PATH} "
                            f"(cvmi_desc_size({kind})); rebuild the library or update the binding to include/cvmi355.h")


def check(rc, what=""):
    if rc != 0:
        msg = load().cvmi_last_error().decode("utf-8", "replace")
        raise CvmiError(f"{what or 'cvmi call'} failed: {msg}")
