"""ctypes binding of libupnerf_hip.so (C ABI: include/upnerf_hip.h).

The product path has no CPU or PyTorch fallback: if the shared library is missing or does not export every
entry point, importing this module raises, loudly.  Build it with `python __graft_entry__.py` (or
`make -C upnerf_amd/csrc`)."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UPNERF_LIB") or os.path.join(_HERE, "libupnerf_hip.so")  # UPNERF_LIB: diagnostic builds only
MAX_D = 8
TILE_ROWS, X0, AUXK, CK = int(os.environ.get("UPNERF_TILE_ROWS", 64)), 64, 80, 16  # env: diagnostic builds only

_fp = C.c_void_p


class Layout(C.Structure):
    _fields_ = [("W", C.c_int32), ("D", C.c_int32), ("skip", C.c_int32),
                ("w", C.c_int32 * MAX_D), ("b", C.c_int32 * MAX_D),
                ("we", C.c_int32), ("be", C.c_int32), ("wsig", C.c_int32), ("bsig", C.c_int32),
                ("wc1", C.c_int32), ("bc1", C.c_int32), ("wc2", C.c_int32), ("bc2", C.c_int32),
                ("wcsig", C.c_int32), ("bcsig", C.c_int32), ("wr1", C.c_int32), ("br1", C.c_int32),
                ("wr2", C.c_int32), ("br2", C.c_int32), ("total", C.c_int32),
                ("t_w", C.c_int32 * MAX_D), ("t_skipx", C.c_int32), ("t_we", C.c_int32), ("t_head", C.c_int32),
                ("t_wc2", C.c_int32), ("t_total", C.c_int32)]


class FieldFwdArgs(C.Structure):
    _fields_ = [("R", C.c_int32), ("S", C.c_int32), ("use_cand", C.c_int32), ("use_rgb", C.c_int32),
                ("rays_o", _fp), ("rays_d", _fp), ("z", _fp), ("c_rows", _fp), ("aux", _fp),
                ("wk_xyz", C.c_float * 10), ("P", _fp),
                ("sigma_s", _fp), ("sigma_c", _fp), ("rgb", _fp),
                ("x0", _fp), ("h", _fp), ("hmask", _fp), ("amax", _fp), ("e", _fp), ("g1", _fp), ("g2", _fp), ("r1", _fp),
                ("P16", _fp), ("wexp", _fp), ("wk_xyz_dev", _fp), ("planes", C.c_int32), ("tile_rows", C.c_int32), ("wnorm", _fp),
                ("h16", _fp), ("hexp", _fp), ("h_last_only", C.c_int32), ("x0f", _fp), ("e16", _fp), ("eexp", _fp),
                ("g2_16", _fp), ("g2exp", _fp), ("r1_16", _fp), ("r1exp", _fp), ("g1_16", _fp), ("g1exp", _fp), ("h_lo8", _fp),
                ("rows_capacity", C.c_int64)]


class CompositeFwdArgs(C.Structure):
    _fields_ = [("R", C.c_int32), ("S", C.c_int32), ("W", C.c_int32), ("mode", C.c_int32),
                ("z", _fp), ("sigma_s", _fp), ("sigma_c", _fp), ("rgb", _fp), ("has_rgb", C.c_int32),
                ("e", _fp), ("g2", _fp),
                ("w_all", _fp), ("w_sj", _fp), ("w_cj", _fp), ("w_s", _fp),
                ("E_s", _fp), ("G_c", _fp), ("sum_sfeat", _fp), ("t_weight", _fp), ("c_depth", _fp),
                ("s_depth", _fp), ("rgb_map", _fp), ("e16", _fp), ("eexp", _fp), ("g2_16", _fp), ("g2exp", _fp),
                ("rgb_joint_map", _fp)]


class CompositeBwdArgs(C.Structure):
    _fields_ = [("R", C.c_int32), ("S", C.c_int32), ("W", C.c_int32), ("mode", C.c_int32), ("has_rgb", C.c_int32),
                ("z", _fp), ("sigma_s", _fp), ("sigma_c", _fp), ("rgb", _fp), ("e", _fp), ("g2", _fp),
                ("w_all", _fp), ("w_sj", _fp), ("w_cj", _fp), ("w_s", _fp),
                ("g_E_s", _fp), ("g_G_c", _fp), ("g_sum_sfeat", _fp), ("g_t_weight", _fp), ("g_c_depth", _fp),
                ("g_s_depth", _fp), ("g_rgb_map", _fp), ("g_w_all", _fp), ("g_w_s", _fp),
                ("d_sigma_s", _fp), ("d_sigma_c", _fp), ("d_rgb", _fp), ("e16", _fp), ("eexp", _fp), ("g2_16", _fp), ("g2exp", _fp),
                ("g_rgb_joint_map", _fp)]


class FieldBwdArgs(C.Structure):
    _fields_ = [("R", C.c_int32), ("S", C.c_int32), ("use_cand", C.c_int32), ("use_rgb", C.c_int32),
                ("need_dxyz", C.c_int32),
                ("PT", _fp), ("P", _fp),
                ("d_sigma_s", _fp), ("d_sigma_c", _fp), ("d_rgb", _fp),
                ("sigma_s", _fp), ("sigma_c", _fp), ("rgb", _fp),
                ("w_feat_s", _fp), ("w_cj", _fp), ("g_E_s", _fp), ("g_G_c", _fp),
                ("x0", _fp), ("h", _fp), ("g1", _fp), ("g2", _fp), ("r1", _fp), ("hmask", _fp), ("gmax", _fp),
                ("gz_h", _fp), ("gz_e", _fp), ("gz_g1", _fp), ("gz_g2", _fp), ("gz_r1", _fp),
                ("dpre_sig_s", _fp), ("dpre_sig_c", _fp), ("dpre_rgb", _fp), ("dxyz", _fp),
                ("PT16", _fp), ("wexp", _fp), ("planes", C.c_int32), ("tile_rows", C.c_int32), ("gz16", _fp), ("gzexp", _fp), ("xs", _fp), ("tile_part", _fp), ("gz_rg_ld", C.c_int32), ("reserved_", C.c_int32), ("wnorm", _fp),
                ("gz_rg16", _fp), ("gzrgexp", _fp), ("gz_g2_16", _fp), ("gzg2exp", _fp), ("gz_lo8", _fp),
                ("rows_capacity", C.c_int64)]


class LossArgs(C.Structure):
    _fields_ = [("R", C.c_int32), ("F", C.c_int32), ("fine", C.c_int32), ("has_tw", C.c_int32),
                ("sched", C.c_float), ("depth_mult", C.c_float), ("alpha_reg", C.c_float), ("near", C.c_float),
                ("far", C.c_float),
                ("depth_direct", _fp), ("inv_depth", _fp), ("depth_scale_rows", _fp),
                ("s_depth_c", _fp), ("s_depth_f", _fp), ("t_weight_c", _fp), ("t_weight_f", _fp),
                ("feat_c", _fp), ("feat_f", _fp), ("feat_gt", _fp),
                ("rgb_c", _fp), ("rgb_f", _fp), ("rgb_gt", _fp), ("beta", _fp), ("alpha", _fp), ("sched_dev", _fp),
                ("term_mask", C.c_int32), ("reserved_", C.c_int32), ("total", _fp), ("g_total", _fp)]


class LossGrads(C.Structure):
    _fields_ = [("d_depth_scale_rows", _fp), ("d_depth", _fp), ("d_s_depth_c", _fp), ("d_s_depth_f", _fp),
                ("d_feat_c", _fp), ("d_feat_f", _fp), ("d_rgb_c", _fp), ("d_rgb_f", _fp), ("d_beta", _fp),
                ("d_alpha", _fp)]


class FragDesc(C.Structure):
    _fields_ = [("src_off", C.c_int32), ("src_ld", C.c_int32), ("transpose", C.c_int32), ("rows", C.c_int32),
                ("cols", C.c_int32), ("dst_off", C.c_int32), ("dst_kp", C.c_int32), ("dst_k0", C.c_int32)]


class PackDesc(C.Structure):
    _fields_ = [("ptr", _fp), ("rows", C.c_int32), ("cols", C.c_int32), ("src_ld", C.c_int32), ("dst_off", C.c_int32),
                ("dst_ld", C.c_int32), ("accumulate", C.c_int32)]


class TransientArgs(C.Structure):
    _fields_ = ([("R", C.c_int32), ("beta_min", C.c_float), ("feat", _fp), ("t_emb", _fp)]
                + [(n, _fp) for n in ("w0", "b0", "w1", "b1", "w2", "b2", "w3", "b3", "wf", "bf", "wt", "bt", "wa", "ba", "wb", "bb",
                                      "wr", "br", "h", "e", "t", "alpha", "rgb", "beta", "spre")])


class TransientGrads(C.Structure):
    _fields_ = [(n, _fp) for n in ("d_alpha", "d_rgb", "d_beta", "dz_heads", "gz_t", "gz_e", "gz_h", "g_temb", "g_feat")]


class AdamDesc(C.Structure):
    _fields_ = [("g", _fp), ("off", C.c_int32), ("n", C.c_int32)]


MAX_ADAM_DESC = 96


class WgradPending(C.Structure):
    _fields_ = [("slabs", _fp), ("bslabs", _fp), ("dW", _fp), ("db", _fp), ("N", C.c_int32), ("K", C.c_int32), ("TN", C.c_int32),
                ("TK", C.c_int32), ("nsplit", C.c_int32), ("ldo", C.c_int32), ("rblocks", C.c_int32), ("n2", C.c_int32),
                ("dW2", _fp), ("db2", _fp), ("ldo2", C.c_int32), ("pad", C.c_int32), ("vslabs", _fp), ("dv", _fp), ("dbv", _fp)]


class WgradGroup(C.Structure):
    _fields_ = [("A", _fp), ("B", _fp), ("dW", _fp), ("db", _fp), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldo", C.c_int32)]


MAX_WGRAD_GROUPS = 32
TILE_PART_STRIDE = 1288  # UPNERF_TILE_PART_STRIDE
RR_PART_STRIDE = 512  # UPNERF_RR_PART_STRIDE


class EmbedGroup(C.Structure):
    _fields_ = [("g", _fp), ("out", _fp), ("dim", C.c_int32)]


class EmbedRowsGroup(C.Structure):
    _fields_ = [("table", _fp), ("rows", _fp), ("dim", C.c_int32)]


MAX_EMBED_GROUPS = 8


class GatherRaysArgs(C.Structure):
    _fields_ = [("R", C.c_int32), ("h", C.c_int32), ("C", C.c_int32), ("idx", _fp),
                ("all_ray_infos", _fp), ("all_directions", _fp), ("all_rgbs", _fp), ("all_pxl_coords", _fp),
                ("all_inv_depths", _fp), ("feat_maps", _fp), ("poses", _fp),
                ("ray_infos", _fp), ("directions", _fp), ("img_idx", _fp), ("c2w", _fp), ("rgbs", _fp), ("feats", _fp),
                ("inv_depths", _fp)]


class Frag16Desc(C.Structure):
    _fields_ = FragDesc._fields_ + [("exp_id", C.c_int32)]


class AddPair(C.Structure):
    _fields_ = [("a", _fp), ("b", _fp), ("out", _fp), ("n", C.c_int32)]


MAX_ADD_PAIRS = 8


class Rng(C.Structure):
    """upnerf_rng: key of the uniform draws a kernel generates itself."""
    _fields_ = [("seed", C.c_uint64), ("step", C.c_int32), ("row0", C.c_int32), ("row_stride", C.c_int32), ("step_dev", _fp)]


_i, _f, _p = C.c_int, C.c_float, C.c_void_p
_SIGNATURES = {
    "upnerf_abi_version": [],
    "upnerf_pose_rays_fwd": [_i, _p, _p, _p, _p, _p, _p],
    "upnerf_pose_rays_bwd": [_i, _p, _p, _p, _p, _p, _p, _p],
    "upnerf_sample_coarse": [_i, _i, _p, _p, _p, _f, _i, _p, _p],
    "upnerf_uniform_keyed": [_i, _i, C.c_uint64, _i, _p, _i, _i, _i, _p, _p],
    "upnerf_sample_pdf": [_i, _i, _p, _p, _p, _i, _i, _p, _i, _p],
    "upnerf_sort_rows": [_i, _i, _p, _p],
    "upnerf_sample_coarse_keyed": [_i, _i, _p, _p, C.POINTER(Rng), _f, _i, _p, _p],
    # (R, Nc, z, w_a, n_a, col_a, u_a, draw_a, w_b, n_b, col_b, u_b, draw_b, u_rows, rng, zf, stream)
    "upnerf_resample_sort": [_i, _i, _p, _p, _i, _i, _p, _i, _p, _i, _i, _p, _i, _i, C.POINTER(Rng), _p, _p],
    "upnerf_ray_aux": [_i, _p, _p, C.POINTER(C.c_float), _p, _p, _p],
    "upnerf_field_fwd": [C.POINTER(Layout), C.POINTER(FieldFwdArgs), _p],
    "upnerf_composite_fwd": [C.POINTER(CompositeFwdArgs), _p],
    "upnerf_composite_bwd": [C.POINTER(CompositeBwdArgs), _p],
    "upnerf_field_bwd": [C.POINTER(Layout), C.POINTER(FieldBwdArgs), _p],
    "upnerf_field_fwd_f16x3": [C.POINTER(Layout), C.POINTER(FieldFwdArgs), _p],
    "upnerf_field_bwd_f16x3": [C.POINTER(Layout), C.POINTER(FieldBwdArgs), _p],
    "upnerf_frag16": [_p, _p, _p, C.POINTER(Frag16Desc), _i, C.POINTER(Frag16Desc), _i, _p, _p, _i, _i, _p, _p],
    "upnerf_wgrad": [_i, _p, _i, _i, _p, _i, _i, _p, _i, _p, _p, _i, _p],
    "upnerf_wgrad_f16x3": [_i, _p, _i, _i, _p, _i, _i, _p, _i, _p, _p, _i, _p, _p, _i, _p],
    "upnerf_wgrad_f16x3_chain": [_i, _p, _i, _i, _p, _i, _i, _p, _i, _p, _p, _i, _p, _p, _i, _p, _p],
    "upnerf_wgrad_f16x3_chain2": [_i, _p, _i, _i, _p, _i, _i, _p, _i, _p, _i, _p, _i, _p, _p, _i, _p, _p, _i, _p, _p],
    # (M, A, lda, N, B, ldb, K, dW, ldo, db, v, dv, dbv, slabs, nsplit, expo_a, expo_b, planes, pending, stream)
    "upnerf_wgrad_f16x3_chain_v": [_i, _p, _i, _i, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _p],
    # (M, A16, aexp, B16, bexp, dW, ldo, db, v, dv, dbv, slabs, nsplit, expo_a, expo_b, pending, stream)
    "upnerf_wgrad_f16p_chain_v": [_i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p],
    "upnerf_wgrad_finish": [_p, _p],
    "upnerf_transient_fwd": [_p, _p],
    "upnerf_transient_bwd": [_p, _p, _p],
    "upnerf_wgrad_f16p": [_i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _p, _i, _p, _p, _i, _p, _p, _p],
    "upnerf_wgrad_f16p_chain": [_i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _p, _i, _p, _i, _p, _i, _p, _p, _i, _p, _p,
                                C.POINTER(WgradPending), _p],
    "upnerf_wgrad_f24p_chain": [_i, _p, _p, _i, _p, _i, _p, _p, _i, _p, _i, _i, _p, _i, _p, _p, _i, _p, _p, C.POINTER(WgradPending), _p],
    # (M, A16, Alo16, aexp, B16, Blo16, bexp, dW, ldo, db, slabs, nsplit, expo_a, expo_b, pending, stream)
    "upnerf_wgrad_planes_chain": [_i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _p, C.POINTER(WgradPending), _p],
    "upnerf_wgrad_grouped_scratch": [C.POINTER(WgradGroup), _i, _i],
    "upnerf_wgrad_grouped": [C.POINTER(WgradGroup), _i, _p, _i, _p],
    "upnerf_vec_wgrad": [_i, _p, _i, _i, _p, _i, _i, _p, _p, _p, _i, _p],
    "upnerf_matvec": [_i, _i, _p, _i, _p, _p, _p, _i, _p],
    "upnerf_matvec_rank1": [_i, _i, _p, _i, _p, _p, _p, _i, _p, _p],
    "upnerf_add_pairs": [C.POINTER(AddPair), _i, _p],
    "upnerf_zero": [_p, C.c_longlong, _p],
    "upnerf_vec_wgrad_frag16": [_i, _p, _i, _i, _p, _p, _i, _p, _p, _p, _i, _p],
    "upnerf_ray_sum": [_i, _i, _p, _i, _p, _p],
    "upnerf_ray_part_finish": [_i, _i, _p, _p, _p, _p],
    "upnerf_tile_part_finish": [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "upnerf_ray_geom_bwd": [_i, _i, _p, _p, _p, _p, _p],
    "upnerf_pack": [_p, C.POINTER(PackDesc), _i, _i, _p],
    "upnerf_gather_rays": [C.POINTER(GatherRaysArgs), _p],
    "upnerf_embed_bwd": [_i, _i, _i, _p, _p, _p, _p],
    "upnerf_embed_bwd_grouped": [_i, _i, _p, C.POINTER(EmbedGroup), _i, _p],
    "upnerf_embed_fwd_grouped": [_i, _i, _p, C.POINTER(EmbedRowsGroup), _i, _p],
    "upnerf_linear": [_i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _i, _p],
    "upnerf_loss_fwd": [C.POINTER(LossArgs), _p, _p, _p, _p],
    "upnerf_loss_bwd": [C.POINTER(LossArgs), _p, C.POINTER(LossGrads), _p],
    "upnerf_frag_copy": [_p, _p, C.POINTER(FragDesc), _i, _p],
    "upnerf_adam": [C.c_int64, _p, _p, _p, _p, _f, _f, _f, _f, _f, _p, _p],
    "upnerf_adam_gather": [_p, _p, _p, _p, _i, _f, _f, _f, _f, _f, _p, _p],
    "upnerf_set_scalars": [_p, _i, C.POINTER(C.c_float), _p],
    "upnerf_scale_exponents": [_p, _i, _p, _p],
}
MAX_SCALARS = 96
EXPORTS = tuple(_SIGNATURES)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            "(there is no CPU/PyTorch fallback for the render_rays hot path).")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it")
        fn.argtypes = argtypes
        fn.restype = C.c_int
    return lib


lib = _load()
ABI_VERSION = 10
if lib.upnerf_abi_version() != ABI_VERSION:
    raise ImportError("libupnerf_hip.so ABI version mismatch; rebuild it")


_PTR_DTYPES = (torch.float32, torch.int64, torch.int32, torch.float16, torch.uint8)


def ptr(t):
    """Device pointer of a contiguous fp32/int64 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    # explicit raises, not asserts: this is the only guard between Python and raw device pointers (python -O keeps it)
    if not t.is_cuda:
        raise RuntimeError("libupnerf_hip operates on device memory only (got a CPU tensor)")
    if not t.is_contiguous():
        raise RuntimeError("non-contiguous tensor handed to the HIP path")
    if t.dtype not in _PTR_DTYPES:
        raise TypeError(f"unsupported dtype {t.dtype} handed to the HIP path")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


CALLS = [0]  # C-ABI calls checked so far (bench.py reports calls per step)


def check(rc: int, what: str):
    CALLS[0] += 1
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported shape"}.get(rc, f"hipError_t {rc}")
        raise RuntimeError(f"{what} failed: {kind}")
