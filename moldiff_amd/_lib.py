"""ctypes binding of libmoldiff_hip.so (declared in include/moldiff_hip.h) + thin torch-tensor wrappers.

There is deliberately NO fallback: if the shared library is missing or a call fails, a RuntimeError is
raised.  Tensors are only used for device memory / streams; every computation happens in the HIP kernels.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmoldiff_hip.so')

MDX_KIND_MOLDIFF, MDX_KIND_BONDPRED, MDX_KIND_NET = 0, 1, 2

EXPORTS = [
    'mdx_last_error', 'mdx_version', 'mdx_device_count',
    'mdx_model_create', 'mdx_model_destroy', 'mdx_model_set_param', 'mdx_model_finalize',
    'mdx_model_set_matrix_path', 'mdx_model_get_matrix_path', 'mdx_model_set_smear_start', 'mdx_gauss_posterior',
    'mdx_graph_create', 'mdx_graph_destroy', 'mdx_graph_plan_host', 'mdx_workspace_bytes',
    'mdx_net_forward', 'mdx_node_block', 'mdx_edge_block', 'mdx_bond_ffn', 'mdx_pos_update', 'mdx_segment_sum',
    'mdx_moldiff_forward', 'mdx_sample_step', 'mdx_sample_step_full', 'mdx_bondpred_forward', 'mdx_bondpred_backward', 'mdx_bondpred_tape_bytes',
    'mdx_pos_posterior', 'mdx_cat_posterior', 'mdx_gumbel_argmax', 'mdx_prior_draw', 'mdx_noise',
    'mdx_guidance_uncertainty_grad', 'mdx_add_inplace', 'mdx_decode_output',
    'mdx_profile_enable', 'mdx_profile_read', 'mdx_profile_kernel_name',
    'mdx_op_sgemm_nt', 'mdx_op_sgemm_tn', 'mdx_op_hgemm_nt', 'mdx_op_hgemm_tn', 'mdx_op_xgemm_nt', 'mdx_op_xgemm_tn', 'mdx_op_amp_adamw',
    'mdx_op_xgemm_nt_t', 'mdx_op_xgemm_nt_ln_t', 'mdx_op_xgemm_nt_ln_supported', 'mdx_op_xgemm_tn_t', 'mdx_op_ln_relu_fwd_t', 'mdx_op_ln_relu_bwd_t', 'mdx_op_ew_fwd_t', 'mdx_op_ew_bwd_t',
    'mdx_op_gather_rows_t', 'mdx_op_segsum_rows_t', 'mdx_op_mul_gather_fwd_t', 'mdx_op_mul_gather_bwd_t',
    'mdx_op_wgrad_layout', 'mdx_op_wgrad_plan', 'mdx_op_wgrad_grouped', 'mdx_op_ln_relu_bwd_rows', 'mdx_op_reduce_deferred', 'mdx_op_transpose', 'mdx_op_transpose_batch', 'mdx_op_linear_rows', 'mdx_op_linear_rows_ws', 'mdx_op_linear_rows_supported', 'mdx_op_colreduce', 'mdx_op_ln_relu_fwd', 'mdx_op_ln_relu_bwd', 'mdx_op_ln_relu_bwd_ws',
    'mdx_op_ew_fwd', 'mdx_op_ew_bwd', 'mdx_op_gather_rows', 'mdx_op_segsum_rows', 'mdx_op_mul_gather_fwd', 'mdx_op_mul_gather_bwd', 'mdx_op_edge_geom_fwd', 'mdx_op_edge_geom_bwd',
    'mdx_op_smear_fwd', 'mdx_op_smear_bwd', 'mdx_op_force_fwd', 'mdx_op_force_bwd', 'mdx_op_sumsq', 'mdx_op_adamw',
    'mdx_op_bondffn_fwd', 'mdx_op_bondffn_bwd', 'mdx_op_bondffn_workgroups', 'mdx_op_bondffn_lnp_floats',
    'mdx_op_edge_tail_fwd', 'mdx_op_edge_tail_bwd', 'mdx_op_edge_tail_lnp_floats',
    'mdx_op_posffn_fwd', 'mdx_op_posffn_bwd', 'mdx_op_posffn_lnp_floats',
    'mdx_op_cat_loss', 'mdx_op_cat_add_noise', 'mdx_op_ln_relu_bwd_r1_t', 'mdx_op_sum_n', 'mdx_op_plan_flip', 'mdx_op_pack_a', 'mdx_op_nodemsg_fwd', 'mdx_op_nodemsg_bwd', 'mdx_op_nodemsg_lnp_floats',
]


class MdxTables(ctypes.Structure):   # == struct mdx_tables
    _fields_ = [(n, c_void_p) for n in ('pos_coef_x0', 'pos_coef_xt', 'pos_std', 'node_q_mats', 'node_qT_onestep', 'edge_q_mats',
                                        'edge_qT_onestep')]


class MdxState(ctypes.Structure):    # == struct mdx_state
    _fields_ = [(n, c_void_p) for n in ('h_node', 'pos', 'h_halfedge', 'log_node', 'log_halfedge')]


class MdxGuidance(ctypes.Structure):   # == struct mdx_guidance
    _fields_ = [('predictor', c_void_p), ('scale', c_float), ('tape', c_void_p), ('tape_bytes', c_size_t), ('ws2', c_void_p),
                ('ws2_bytes', c_size_t), ('logits', c_void_p), ('glogits', c_void_p), ('delta', c_void_p),
                ('side_stream', c_void_p)]


class MdxStepNoise(ctypes.Structure):  # == struct mdx_step_noise
    _fields_ = [('seed', c_uint64), ('draw', c_int32), ('eps_pos', c_void_p), ('u_node', c_void_p), ('u_halfedge', c_void_p)]


class MdxBondFfnArgs(ctypes.Structure):   # == mdx_bondffn_args
    _fields_ = [('X', c_void_p), ('ldx', c_int64), ('Wb', c_void_p), ('ldwb', c_int64),
                ('Wi1', c_void_p), ('ldwi1', c_int64), ('bi1', c_void_p), ('g1', c_void_p), ('be1', c_void_p),
                ('Wi2', c_void_p), ('ldwi2', c_int64), ('bi2', c_void_p),
                ('Wg1', c_void_p), ('ldwg1', c_int64), ('bg1', c_void_p), ('gg', c_void_p), ('gbe', c_void_p),
                ('Wt', c_void_p), ('ldwt', c_int64), ('Wg2', c_void_p), ('ldwg2', c_int64), ('bg2', c_void_p),
                ('NL', c_void_p), ('ldnl', c_int64), ('GN', c_void_p), ('ldgn', c_int64), ('idx', c_void_p), ('te', c_void_p),
                ('prod', c_void_p), ('pre1', c_void_p), ('post1', c_void_p), ('inter', c_void_p), ('gpre', c_void_p), ('gpost', c_void_p),
                ('gate', c_void_p), ('out', c_void_p), ('E', c_int64)]


class MdxBondFfnBwdArgs(ctypes.Structure):   # == mdx_bondffn_bwd_args
    _fields_ = [('f', MdxBondFfnArgs), ('gS', c_void_p), ('ldgs', c_int64), ('oidx', c_void_p),
                ('g_inter', c_void_p), ('g_gate', c_void_p), ('g_pre1', c_void_p), ('g_bf', c_void_p), ('g_nl', c_void_p),
                ('g_gpre', c_void_p), ('g_x', c_void_p), ('lnp', c_void_p)]


class MdxEdgeTailArgs(ctypes.Structure):   # == mdx_edge_tail_args
    _fields_ = [('H', c_void_p), ('ldh', c_int64), ('BL', c_void_p), ('ldbl', c_int64), ('BR', c_void_p), ('ldbr', c_int64),
                ('il', c_void_p), ('ir', c_void_p), ('Ws', c_void_p), ('ldws', c_int64), ('bs', c_void_p), ('lng', c_void_p), ('lnb', c_void_p),
                ('Wo', c_void_p), ('ldwo', c_int64), ('bo', c_void_p), ('pre', c_void_p), ('post', c_void_p), ('out', c_void_p), ('E', c_int64)]


class MdxEdgeTailBwdArgs(ctypes.Structure):   # == mdx_edge_tail_bwd_args
    _fields_ = [('f', MdxEdgeTailArgs), ('g_out', c_void_p), ('ldg', c_int64), ('g_pre', c_void_p), ('g_h', c_void_p), ('lnp', c_void_p)]


class MdxPosFfnArgs(ctypes.Structure):   # == mdx_posffn_args
    _fields_ = [('X', c_void_p), ('ldx', c_int64), ('LF', c_void_p), ('ldlf', c_int64), ('RF', c_void_p), ('ldrf', c_int64),
                ('il', c_void_p), ('ir', c_void_p), ('te', c_void_p), ('Wb', c_void_p), ('ldwb', c_int64), ('Wn', c_void_p), ('ldwn', c_int64),
                ('Wg1x', c_void_p), ('ldwg1x', c_int64), ('Wg1a', c_void_p), ('ldwg1a', c_int64), ('Wt', c_void_p), ('ldwt', c_int64),
                ('bg1', c_void_p), ('gg', c_void_p), ('gbe', c_void_p), ('Wg2', c_void_p), ('bg2', c_void_p),
                ('a', c_void_p), ('prod', c_void_p), ('gpre', c_void_p), ('gpost', c_void_p), ('gate', c_void_p), ('E', c_int64)]


class MdxPosFfnBwdArgs(ctypes.Structure):   # == mdx_posffn_bwd_args
    _fields_ = [('f', MdxPosFfnArgs), ('g_prod', c_void_p), ('ldgp', c_int64), ('g_gate', c_void_p), ('g_bf', c_void_p), ('g_nf', c_void_p),
                ('g_gpre', c_void_p), ('g_x', c_void_p), ('g_lf', c_void_p), ('g_rf', c_void_p), ('lnp', c_void_p)]


class MdxPackJob(ctypes.Structure):   # == mdx_pack_job
    _fields_ = [('W', c_void_p), ('ld', c_int64), ('n_out', c_int32), ('n_in', c_int32), ('perm', c_int32), ('trans', c_int32), ('out', c_void_p)]


class MdxPackJobs(ctypes.Structure):   # == mdx_pack_jobs
    _fields_ = [('job', MdxPackJob * 10), ('n', c_int32)]


class MdxNodeMsgArgs(ctypes.Structure):   # == mdx_nodemsg_args
    _fields_ = ([('X', c_void_p), ('ldx', c_int64), ('HN', c_void_p), ('ldhn', c_int64), ('PN', c_void_p), ('ldpn', c_int64), ('col', c_void_p)] +
                [(n, c_void_p) for n in ('pk_w1e', 'pk_w2e', 'pk_wm', 'pk_wg1', 'pk_wg2', 'b1e', 'lng_e', 'lnb_e', 'b2e', 'bm', 'bg1', 'lng_g',
                                         'lnb_g', 'bg2', 'he_pre', 'he_post', 'he', 'p', 'm0', 'g_pre', 'g_post', 'gt', 'msg')] + [('E', c_int64)])


class MdxNodeMsgBwdArgs(ctypes.Structure):   # == mdx_nodemsg_bwd_args
    _fields_ = ([('f', MdxNodeMsgArgs), ('gA', c_void_p), ('ldga', c_int64), ('row', c_void_p)] +
                [(n, c_void_p) for n in ('pk_wg2t', 'pk_wg1t', 'pk_wmt', 'pk_w2et', 'pk_w1et', 'g_m0', 'g_gt', 'g_gpre', 'g_hne', 'g_he', 'g_pre',
                                         'g_x', 'lnp')])


class MdxConfig(ctypes.Structure):
    _fields_ = [('kind', c_int32), ('node_dim', c_int32), ('edge_dim', c_int32), ('num_blocks', c_int32),
                ('cutoff', c_float), ('num_gaussians', c_int32), ('update_pos', c_int32), ('time_dim', c_int32),
                ('num_timesteps', c_int32), ('num_node_types', c_int32), ('num_edge_types', c_int32)]


_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                               f'(or `make -C moldiff_amd/csrc`). moldiff_amd has no CPU/PyTorch fallback.')
        L = ctypes.CDLL(LIB_PATH)
        L.mdx_last_error.restype = c_char_p
        L.mdx_workspace_bytes.restype = c_size_t
        L.mdx_workspace_bytes.argtypes = [c_int64, c_int64]
        L.mdx_model_create.argtypes = [POINTER(MdxConfig), POINTER(c_void_p)]
        L.mdx_model_destroy.argtypes = [c_void_p]
        L.mdx_model_set_param.argtypes = [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int32]
        L.mdx_model_finalize.argtypes = [c_void_p]
        L.mdx_model_set_matrix_path.argtypes = [c_void_p, c_int32]
        L.mdx_model_set_smear_start.argtypes = [c_void_p, c_float]
        L.mdx_model_get_matrix_path.argtypes = [c_void_p, POINTER(c_int32)]
        L.mdx_graph_create.argtypes = [c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, POINTER(c_void_p)]
        L.mdx_graph_destroy.argtypes = [c_void_p]
        L.mdx_graph_plan_host.argtypes = [c_int64, c_int64] + [c_void_p] * 7
        L.mdx_net_forward.argtypes = [c_void_p] * 10 + [c_void_p, c_size_t, c_void_p]
        L.mdx_node_block.argtypes = [c_void_p, c_void_p, c_int32] + [c_void_p] * 4 + [c_void_p, c_size_t, c_void_p]
        L.mdx_edge_block.argtypes = [c_void_p, c_void_p, c_int32] + [c_void_p] * 4 + [c_void_p, c_size_t, c_void_p]
        L.mdx_bond_ffn.argtypes = [c_void_p, c_void_p, c_int32, c_int32] + [c_void_p] * 4 + [c_void_p, c_size_t, c_void_p]
        L.mdx_pos_update.argtypes = [c_void_p, c_void_p, c_int32] + [c_void_p] * 6 + [c_void_p, c_size_t, c_void_p]
        L.mdx_segment_sum.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]
        L.mdx_moldiff_forward.argtypes = [c_void_p] * 10 + [c_void_p, c_size_t, c_void_p]
        L.mdx_sample_step.argtypes = [c_void_p, c_void_p, POINTER(MdxTables), c_void_p, c_void_p, c_void_p, POINTER(MdxState),
                                      POINTER(MdxState)] + [c_void_p] * 6 + [c_void_p, c_size_t, c_void_p]
        L.mdx_sample_step_full.argtypes = [c_void_p, c_void_p, POINTER(MdxTables), c_int32, c_void_p, c_void_p, POINTER(MdxState),
                                           POINTER(MdxState), c_void_p, c_void_p, c_void_p, POINTER(MdxStepNoise), c_void_p,
                                           c_void_p, c_void_p, POINTER(MdxGuidance), c_void_p, c_size_t, c_void_p]
        L.mdx_bondpred_forward.argtypes = [c_void_p] * 6 + [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]
        L.mdx_bondpred_backward.argtypes = [c_void_p] * 4 + [c_float, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t,
                                                            c_void_p]
        L.mdx_bondpred_tape_bytes.restype = c_size_t
        L.mdx_bondpred_tape_bytes.argtypes = [c_int64, c_int64, c_int32]
        L.mdx_pos_posterior.argtypes = [c_void_p] * 8 + [c_int64, c_void_p, c_void_p]
        L.mdx_gauss_posterior.argtypes = [c_void_p] * 8 + [c_int64, c_int32, c_void_p, c_void_p]
        L.mdx_cat_posterior.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p,
                                        c_void_p, c_int64, c_void_p, c_void_p]
        L.mdx_gumbel_argmax.argtypes = [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p]
        L.mdx_op_cat_add_noise.argtypes = [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p,
                                           c_void_p, c_void_p]
        L.mdx_op_cat_loss.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_void_p]
        L.mdx_prior_draw.argtypes = [c_void_p, c_int32, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]
        L.mdx_noise.argtypes = [c_void_p, c_uint64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]
        L.mdx_guidance_uncertainty_grad.argtypes = [c_void_p, c_int32, c_int64, c_void_p, c_void_p]
        L.mdx_add_inplace.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
        L.mdx_decode_output.argtypes = ([c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32] +
                                        [c_void_p] * 8 + [c_void_p, c_size_t, c_void_p])
        L.mdx_device_count.argtypes = [POINTER(c_int)]
        L.mdx_profile_enable.argtypes = [c_int32]
        L.mdx_profile_read.argtypes = [c_int32, POINTER(c_int64), POINTER(ctypes.c_double)]
        L.mdx_profile_kernel_name.argtypes = [c_int32]
        L.mdx_profile_kernel_name.restype = c_char_p
        # layer-level training operators
        L.mdx_op_sgemm_nt.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                      c_int64, c_int64, c_int32, c_void_p, c_void_p]
        L.mdx_op_sgemm_tn.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                      c_int32, c_void_p, c_void_p]
        L.mdx_op_hgemm_nt.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                      c_int64, c_int64, c_void_p]
        L.mdx_op_hgemm_tn.argtypes = L.mdx_op_sgemm_tn.argtypes
        L.mdx_op_xgemm_nt.argtypes = L.mdx_op_hgemm_nt.argtypes[:-1] + [c_int32, c_int32, c_void_p]
        L.mdx_op_xgemm_tn.argtypes = L.mdx_op_sgemm_tn.argtypes[:-1] + [c_int32, c_int32, c_void_p]
        L.mdx_op_amp_adamw.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_float,
                                       c_void_p, c_float, c_float, c_int32, c_void_p, c_void_p]
        L.mdx_op_transpose.argtypes = [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]
        L.mdx_op_transpose_batch.argtypes = [c_void_p, c_int64, c_int64, c_void_p]
        L.mdx_op_linear_rows.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                         c_int64, c_int64, c_int64, c_void_p, c_void_p]
        L.mdx_op_linear_rows_ws.argtypes = [c_int64, c_int64]
        L.mdx_op_linear_rows_ws.restype = ctypes.c_size_t
        L.mdx_op_linear_rows_supported.argtypes = [c_int64, c_int64]
        L.mdx_op_colreduce.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]
        L.mdx_op_ln_relu_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
        L.mdx_op_ln_relu_bwd.argtypes = [c_void_p] * 5 + [c_int64, c_int32, c_int32] + [c_void_p] * 4
        L.mdx_op_ln_relu_bwd_ws.restype = c_size_t
        L.mdx_op_ln_relu_bwd_ws.argtypes = [c_int64, c_int32]
        L.mdx_op_ew_fwd.argtypes = [c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
        L.mdx_op_ew_bwd.argtypes = [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
        L.mdx_op_gather_rows.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]
        L.mdx_op_segsum_rows.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]
        L.mdx_op_mul_gather_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]
        L.mdx_op_mul_gather_bwd.argtypes = [c_void_p] * 6 + [c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p]
        L.mdx_op_wgrad_layout.argtypes = [c_int64, c_int64, c_int64, c_int32, c_int32, POINTER(c_int64), POINTER(c_int64)]
        L.mdx_op_wgrad_plan.argtypes = [c_int64, c_int64, c_int64, c_int32, c_int32, c_int64, c_int64, c_int32, POINTER(c_int64)]
        L.mdx_op_wgrad_grouped.argtypes = [c_void_p, c_int32, c_int64, c_int32, c_void_p]
        L.mdx_op_sum_n.argtypes = [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_int32, c_void_p]
        L.mdx_op_plan_flip.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]
        L.mdx_op_ln_relu_bwd_r1_t.argtypes = [c_void_p] * 6 + [c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p]
        L.mdx_op_ln_relu_bwd_rows.argtypes = [c_int64]
        L.mdx_op_ln_relu_bwd_rows.restype = c_int64
        L.mdx_op_reduce_deferred.argtypes = [c_void_p, c_int32, c_int64, c_void_p]
        # half-storage forms: the fp32 signature with an int32 `dt` mask in front of the stream
        for name in ('xgemm_nt', 'xgemm_tn', 'ln_relu_fwd', 'ln_relu_bwd', 'ew_fwd', 'ew_bwd', 'gather_rows', 'segsum_rows', 'mul_gather_fwd',
                     'mul_gather_bwd'):
            base = getattr(L, 'mdx_op_' + name).argtypes
            getattr(L, 'mdx_op_' + name + '_t').argtypes = list(base[:-1]) + [c_int32, c_void_p]
        L.mdx_op_xgemm_nt_ln_supported.argtypes = [c_int64, c_int64, c_int64]
        L.mdx_op_xgemm_nt_ln_t.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                           c_void_p, c_int64, c_void_p, c_int32, c_int64, c_int64, c_int64, c_int32, c_int32, c_int32, c_void_p]
        L.mdx_op_edge_geom_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
        L.mdx_op_edge_geom_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
        L.mdx_op_smear_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float, c_int64, c_void_p, c_void_p]
        L.mdx_op_smear_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float, c_int64, c_void_p, c_void_p, c_void_p]
        L.mdx_op_force_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
        L.mdx_op_force_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
        L.mdx_op_sumsq.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
        L.mdx_op_adamw.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                                   c_int64, c_void_p, c_float, c_void_p]
        L.mdx_op_bondffn_fwd.argtypes = [POINTER(MdxBondFfnArgs), c_void_p]
        L.mdx_op_bondffn_bwd.argtypes = [POINTER(MdxBondFfnBwdArgs), c_void_p]
        L.mdx_op_edge_tail_fwd.argtypes = [POINTER(MdxEdgeTailArgs), c_void_p]
        L.mdx_op_edge_tail_bwd.argtypes = [POINTER(MdxEdgeTailBwdArgs), c_void_p]
        L.mdx_op_posffn_fwd.argtypes = [POINTER(MdxPosFfnArgs), c_void_p]
        L.mdx_op_posffn_bwd.argtypes = [POINTER(MdxPosFfnBwdArgs), c_void_p]
        L.mdx_op_pack_a.argtypes = [POINTER(MdxPackJobs), c_void_p]
        L.mdx_op_nodemsg_fwd.argtypes = [POINTER(MdxNodeMsgArgs), c_void_p]
        L.mdx_op_nodemsg_bwd.argtypes = [POINTER(MdxNodeMsgBwdArgs), c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(f'libmoldiff_hip error {rc}: {lib().mdx_last_error().decode()}')


def _need_gpu(*tensors):
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('moldiff_amd runs on a ROCm device only (got a CPU tensor); there is no CPU fallback. '
                               'The CPU oracle lives in oracle/ and is test infrastructure.')
        # the library allocates and launches on the CURRENT HIP device and stream: tensors of another device would be addressed
        # from the wrong GPU's queue (memory fault or unordered execution), so say so instead
        if cur is None:
            cur = _current_device()
        if t.get_device() != cur:
            raise RuntimeError(f'tensor on {t.device} but the current device is cuda:{cur}: call torch.cuda.set_device(...) first '
                               '(one process per GPU)')


# These two run a few thousand times per training step (every operator call): the raw torch._C entry points cost ~1 us, the
# torch.cuda wrappers (lazy-init check, Stream object construction) ~9 us -- on a step of ~1,500 operators that was 15 % of the
# host time of a step that the host, not the GPU, bounds.
_raw_device = getattr(torch._C, '_cuda_getDevice', None)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _current_device():
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


def ptr(t):
    """address of a tensor's first element as a plain int (None = NULL): every pointer parameter is declared c_void_p in `argtypes`,
    which converts both, and no c_void_p object is built per argument"""
    return t.data_ptr() if t is not None else None


def stream():
    if _raw_stream is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def f32c(t):
    return t.detach().to(torch.float32).contiguous()


def i64c(t):
    return t.detach().to(torch.int64).contiguous()


class Graph:
    """Device CSR plan of one packed batch (edge list stably sorted by (left, right))."""

    def __init__(self, edge_index, batch_node, n_graphs, mol_ids=None):
        ei = edge_index.detach().to('cpu', torch.int64).contiguous()
        bn = batch_node.detach().to('cpu', torch.int64).contiguous()
        self.N, self.E, self.B = int(bn.numel()), int(ei.shape[1]), int(n_graphs)
        self.Eh = self.E // 2
        mids = None
        if mol_ids is not None:
            mids = torch.as_tensor(mol_ids, dtype=torch.int64).contiguous()
        h = c_void_p()
        check(lib().mdx_graph_create(self.N, self.E, ptr(ei), ptr(bn), self.B, ptr(mids), ctypes.byref(h)))
        self.h = h
        self._ws = None

    def workspace(self, device):
        if self._ws is None or self._ws.device != device:
            nbytes = lib().mdx_workspace_bytes(self.N, self.E)
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        off = (-self._ws.data_ptr()) % 256
        return c_void_p(self._ws.data_ptr() + off), c_size_t(self._ws.numel() - off)

    def tape(self, device, num_blocks):
        """Caller-owned tape + backward scratch of the bond predictor (allocated on first guided call)."""
        key = (str(device), num_blocks)
        if getattr(self, '_tape_key', None) != key:
            nbytes = lib().mdx_bondpred_tape_bytes(self.N, self.E, num_blocks)
            self._tape = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
            self._tape_key = key
        off = (-self._tape.data_ptr()) % 256
        return self._tape, c_void_p(self._tape.data_ptr() + off), c_size_t(self._tape.numel() - off)

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                lib().mdx_graph_destroy(self.h)
        except Exception:
            pass


_graph_cache = {}     # identity-keyed, least recently used first; each Graph owns a workspace (hundreds of MB at 256 molecules)
_GRAPH_CACHE_MAX = 3


def _cache_get(key):
    g = _graph_cache.pop(key, None)
    if g is not None:
        _graph_cache[key] = g   # most recently used last
    return g


def _cache_put(key, g):
    _graph_cache[key] = g
    while len(_graph_cache) > _GRAPH_CACHE_MAX:
        _graph_cache.pop(next(iter(_graph_cache)))


def graph_for(edge_index, batch_node, n_graphs=None, mol_ids=None):
    """Small identity-keyed LRU so repeated forward() calls on the same index tensors reuse the plan."""
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, batch_node.data_ptr(),
           batch_node._version, int(batch_node.numel()), n_graphs)
    g = _cache_get(key)
    if g is None:
        if n_graphs is None:
            n_graphs = int(batch_node.max().item()) + 1 if batch_node.numel() else 0
        g = Graph(edge_index, batch_node, n_graphs, mol_ids)
        g._keepalive = (edge_index, batch_node)
        _cache_put(key, g)
    return g


def graph_for_halfedges(halfedge_index, batch_node, n_graphs):
    """Plan of edge_index = cat[halfedge_index, flip(halfedge_index)] keyed on the HALF-edge tensor: callers that rebuild the
    directed list with torch.cat on every call (get_loss, decode_batch) still hit the cache."""
    key = ('half', halfedge_index.data_ptr(), tuple(halfedge_index.shape), halfedge_index._version, batch_node.data_ptr(),
           batch_node._version, int(batch_node.numel()), n_graphs)
    g = _cache_get(key)
    if g is None:
        g = Graph(torch.cat([halfedge_index, halfedge_index.flip(0)], dim=1), batch_node, n_graphs)
        g._keepalive = (halfedge_index, batch_node)
        _cache_put(key, g)
    return g


# Matrix path of the per-edge Linear layers (include/moldiff_hip.h: MDX_MATRIX_*).  'exact_f32' is the default everywhere;
# 'split_f16' is opt-in per module (`module.matrix_path = 'split_f16'`) or process-wide through default_matrix_path().
MATRIX_PATHS = {'exact_f32': 0, 'split_f16': 1}
_default_matrix_path = os.environ.get('MOLDIFF_MATRIX_PATH', 'exact_f32')


class default_matrix_path:
    """`with default_matrix_path('split_f16'):` -- modules whose own `matrix_path` is None follow the process default."""

    def __init__(self, name):
        if name not in MATRIX_PATHS:
            raise ValueError(f'unknown matrix path {name!r} (one of {sorted(MATRIX_PATHS)})')
        self.name = name

    def __enter__(self):
        global _default_matrix_path
        self.prev, _default_matrix_path = _default_matrix_path, self.name
        return self

    def __exit__(self, *exc):
        global _default_matrix_path
        _default_matrix_path = self.prev


class pinned_matrix_path:
    """Run a module's forward on the path a sampler resolved at construction, whatever the process default is by now: the
    module's own `matrix_path` attribute is set for the duration of the block (its _engine() resolves that first)."""

    def __init__(self, module, name):
        self.module, self.name = module, name

    def __enter__(self):
        self.prev = self.module.matrix_path
        self.module.matrix_path = self.name
        return self

    def __exit__(self, *exc):
        self.module.matrix_path = self.prev


def resolve_matrix_path(name):
    name = _default_matrix_path if name is None else name
    if name not in MATRIX_PATHS:
        raise ValueError(f'unknown matrix path {name!r} (one of {sorted(MATRIX_PATHS)})')
    return name


class Model:
    """Packed weights of one MolDiff / BondPredictor / bare NodeEdgeNet on the device."""
    _path = 'exact_f32'

    def use_matrix_path(self, name):
        """Select the matrix path for the following calls (a no-op when it is already selected)."""
        name = resolve_matrix_path(name)
        if name != self._path:
            check(lib().mdx_model_set_matrix_path(self.h, MATRIX_PATHS[name]))
            self._path = name
        return self

    def __init__(self, kind, *, num_blocks, cutoff, update_pos, time_dim=0, num_timesteps=1, num_node_types=1,
                 num_edge_types=1, node_dim=256, edge_dim=64, num_gaussians=16, smear_start=0.0):
        cfg = MdxConfig(kind, node_dim, edge_dim, num_blocks, float(cutoff), num_gaussians, int(bool(update_pos)),
                        time_dim, num_timesteps, num_node_types, num_edge_types)
        h = c_void_p()
        check(lib().mdx_model_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h
        self.cfg = cfg
        if float(smear_start) != 0.0:   # GaussianSmearing(start != 0): models/graph.py:330-333
            check(lib().mdx_model_set_smear_start(h, float(smear_start)))

    def upload(self, state_dict, prefix=''):
        for k, v in state_dict.items():
            if not k.startswith(prefix) or v.dtype not in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
                continue
            t = v.detach().to('cpu', torch.float32).contiguous()
            shape = (c_int64 * max(t.dim(), 1))(*t.shape)
            check(lib().mdx_model_set_param(self.h, k[len(prefix):].encode(), ptr(t), shape, t.dim()))
        check(lib().mdx_model_finalize(self.h))
        # finalize drops a handle back to the exact path when the new weights cannot be held by the split packs (float16 range): the
        # cached name must follow the handle, or a later use_matrix_path('split_f16') would be a silent no-op on the exact path
        got = ctypes.c_int32()
        check(lib().mdx_model_get_matrix_path(self.h, ctypes.byref(got)))
        now = next(n for n, v in MATRIX_PATHS.items() if v == got.value)
        if now != self._path:
            was, self._path = self._path, now
            raise RuntimeError(f"weights re-uploaded to a model on matrix path '{was}' cannot be held by it (|w| beyond the float16 "
                               f"range); the handle is on '{now}' now -- select the path again explicitly if that is intended")

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                lib().mdx_model_destroy(self.h)
        except Exception:
            pass


# ---- transition wrappers (used by moldiff_amd.transition) ------------------------------------------

def pos_posterior(c0, ct, sd, x_t, x_recon, eps, t, batch):
    _need_gpu(x_t, x_recon, eps, t, batch, c0)
    # converted copies are bound to names that live until the launch: a temporary handed straight to ptr() goes back to the
    # caching allocator at once and the next conversion in the same argument list may reuse its block
    x_t, x_recon, eps, t, batch = f32c(x_t), f32c(x_recon), f32c(eps), i64c(t), i64c(batch)
    out = torch.empty_like(x_t)
    if x_t.shape[-1] == 3:
        check(lib().mdx_pos_posterior(ptr(c0), ptr(ct), ptr(sd), ptr(x_t), ptr(x_recon), ptr(eps), ptr(t),
                                      ptr(batch), x_t.shape[0], ptr(out), stream()))
    else:   # class features of the continuous categorical space
        check(lib().mdx_gauss_posterior(ptr(c0), ptr(ct), ptr(sd), ptr(x_t), ptr(x_recon), ptr(eps), ptr(t),
                                        ptr(batch), x_t.shape[0], x_t.shape[-1], ptr(out), stream()))
    return out


def cat_posterior(q_mats, qT, in0, log_vt, t, batch, is_logits=False):
    _need_gpu(in0, log_vt, t, batch, q_mats)
    in0, log_vt, t, batch = f32c(in0), f32c(log_vt), i64c(t), i64c(batch)
    out = torch.empty_like(in0)
    check(lib().mdx_cat_posterior(ptr(q_mats), ptr(qT), in0.shape[1], q_mats.shape[0], ptr(in0), int(is_logits),
                                  ptr(log_vt), ptr(t), ptr(batch), in0.shape[0], ptr(out), stream()))
    return out


_LOG_EPS32 = None


def cat_add_noise(q_mats, v, t, batch, u, K):
    """GeneralCategoricalTransition.add_noise's arithmetic in one launch (csrc cat_add_noise_kernel) -> (onehot, log_vt, log_v0)"""
    global _LOG_EPS32
    _need_gpu(v, t, batch, u, q_mats)
    if _LOG_EPS32 is None:
        _LOG_EPS32 = float(torch.log(torch.tensor([1e-30], dtype=torch.float32))[0])
    v, t, batch, u = i64c(v), i64c(t), i64c(batch), f32c(u)
    n = v.shape[0]
    oh, lvt, lv0 = (torch.empty(n, K, dtype=torch.float32, device=v.device) for _ in range(3))
    check(lib().mdx_op_cat_add_noise(ptr(q_mats), K, q_mats.shape[0], ptr(v), ptr(t), ptr(batch), ptr(u), n, _LOG_EPS32, ptr(oh), ptr(lvt), ptr(lv0),
                                     stream()))
    return oh, lvt, lv0


def prior_draw(init_prob, u, n, cls=None, onehot=None, log_onehot=None, cls8=None):
    """GeneralCategoricalTransition.sample_init (models/transition.py:331-339): Gumbel-max in float64 on the float64 prior logits.
    init_prob: float64 numpy (K); u: (n,K) device tensor, float64 or float32; outputs are written where given."""
    global _LOG_EPS32
    import numpy as np
    _need_gpu(u)
    if u.dtype not in (torch.float32, torch.float64) or not u.is_contiguous():
        u = u.to(torch.float64).contiguous()
    K = int(init_prob.shape[0])
    lg = torch.log(torch.from_numpy(np.asarray(init_prob, dtype=np.float64)) + 1e-30).clamp_min(-32.).contiguous()   # float64, host
    if _LOG_EPS32 is None:
        _LOG_EPS32 = float(torch.log(torch.tensor([1e-30], dtype=torch.float32))[0])
    check(lib().mdx_prior_draw(lg.data_ptr(), K, ptr(u), int(u.dtype == torch.float64), n, ptr(cls), ptr(onehot), ptr(log_onehot),
                               _LOG_EPS32, ptr(cls8), stream()))


def gumbel_argmax(logits, u, want_onehot=False):
    _need_gpu(logits, u)
    logits, u = f32c(logits), f32c(u)
    n, K = logits.shape
    cls = torch.empty(n, dtype=torch.int64, device=logits.device)
    oh = torch.empty(n, K, dtype=torch.float32, device=logits.device) if want_onehot else None
    check(lib().mdx_gumbel_argmax(ptr(logits), ptr(u), K, n, ptr(cls), ptr(oh), stream()))
    return (cls, oh) if want_onehot else cls
