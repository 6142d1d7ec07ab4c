"""ctypes boundary to ``ext/libssnode.so`` (HIP / gfx950 build).

Drop-in for the reference's ``tc_gan/clib.py`` (lines 7-33): same loader
(``numpy.ctypeslib.load_library('libssnode', <pkg>/ext)`` at import time, so a
missing library is an ``OSError`` on import), same ``libssnode`` object with the
same argtypes/restype for the eight reference symbols, plus the additive
batched ABI declared in ``include/ssnode_mi355x.h``.
"""
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_long, c_void_p
import collections
import ctypes
import os

import numpy

# One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 /
# libhsa-runtime64 (same SONAMEs as /opt/rocm's).  If libssnode.so pulled in the
# system copies first, torch's later initialisation would find "No HIP GPUs".
# Loading torch first makes the dynamic linker resolve libssnode's libamdhip64.so.7
# to the copy that is already mapped, so kernels, streams and device pointers are
# shared with torch tensors.  (torch is plumbing here: device memory and streams.)
try:
    import torch  # noqa: F401
except ImportError:      # pure-ctypes use without torch: the system runtime is used
    torch = None

# (SSN_LIBDIR: A/B timing of two builds of the library in one session -- tools/ab_build.sh; never set in production)
libdir = os.environ.get('SSN_LIBDIR') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ext')


def load_library(name):
    return numpy.ctypeslib.load_library(name, libdir)


double_ptr = ctypes.POINTER(ctypes.c_double)

libssnode = load_library('libssnode')

# ---- reference symbols (tc_gan/clib.py:16-33) ---------------------------------
for fun in [libssnode.solve_dynamics_asym_power_euler,
            libssnode.solve_dynamics_asym_linear_euler,
            libssnode.solve_dynamics_asym_tanh_euler]:
    fun.argtypes = [
        c_int, double_ptr, double_ptr, c_double, c_double,
        double_ptr, double_ptr,
        c_double, c_double,
        c_double, c_int, c_double,
        c_double, c_double,
    ]
    fun.restype = ctypes.c_int

for fun in [libssnode.io_pow, libssnode.io_alin, libssnode.io_atanh]:
    fun.argtypes = [c_double] * 6
    fun.restype = c_double

libssnode.rate_to_volt.argtypes = [c_double] * 3
libssnode.rate_to_volt.restype = c_double

libssnode.dot.argtypes = [c_int, double_ptr, double_ptr]
libssnode.dot.restype = c_double


# ---- additive batched ABI (include/ssnode_mi355x.h section 2) ------------------
SSN_IO_POWER, SSN_IO_LINEAR, SSN_IO_TANH = 0, 1, 2
SSN_ERR_BASE = 1000
IO_CODES = {'asym_power': SSN_IO_POWER, 'asym_linear': SSN_IO_LINEAR, 'asym_tanh': SSN_IO_TANH}


class SolverParams(Structure):
    """``ssn_solver_params`` of include/ssnode_mi355x.h."""
    _fields_ = [
        ('io_type', c_int), ('max_iter', c_int),
        ('k', c_double), ('n', c_double),
        ('tau_E', c_double), ('tau_I', c_double),
        ('dt', c_double), ('atol', c_double),
        ('rate_soft_bound', c_double), ('rate_hard_bound', c_double),
    ]


_pp = POINTER(SolverParams)

libssnode.ssn_abi_version.argtypes = []
libssnode.ssn_abi_version.restype = c_int
libssnode.ssn_device_count.argtypes = []
libssnode.ssn_device_count.restype = c_int
libssnode.ssn_last_error.argtypes = []
libssnode.ssn_last_error.restype = c_char_p
libssnode.ssn_solver_fast_path.argtypes = [c_int, c_int, c_int]
libssnode.ssn_solver_fast_path.restype = c_int

_solve_args = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
               c_int, c_int, c_int, _pp, c_void_p]
for _name in ('ssn_solve_batch_f32', 'ssn_solve_batch_f64'):
    getattr(libssnode, _name).argtypes = _solve_args
    getattr(libssnode, _name).restype = c_int
for _name in ('ssn_solve_batch_f32_variant', 'ssn_solve_batch_f64_variant'):
    getattr(libssnode, _name).argtypes = [c_int] + _solve_args
    getattr(libssnode, _name).restype = c_int
for _name in ('ssn_solve_batch_host_f32', 'ssn_solve_batch_host_f64'):
    getattr(libssnode, _name).argtypes = _solve_args[:-1]
    getattr(libssnode, _name).restype = c_int

libssnode.ssn_build_w_f32.argtypes = [c_void_p, POINTER(c_float), POINTER(c_float), POINTER(c_float),
                                      c_void_p, c_int, c_int, c_void_p]
libssnode.ssn_build_w_f64.argtypes = [c_void_p, POINTER(c_double), POINTER(c_double), POINTER(c_double),
                                      c_void_p, c_int, c_int, c_void_p]
libssnode.ssn_stimulus_f32.argtypes = [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p]
libssnode.ssn_stimulus_f64.argtypes = [c_void_p, c_void_p, c_double, c_void_p, c_int, c_int, c_int, c_void_p]
libssnode.ssn_stimulus_amp_f32.argtypes = [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
libssnode.ssn_stimulus_amp_f64.argtypes = [c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
libssnode.ssn_stimulus_hetero_f32.argtypes = [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]
libssnode.ssn_stimulus_hetero_f32.restype = c_int
libssnode.ssn_stimulus_amp_f32.restype = c_int
libssnode.ssn_stimulus_amp_f64.restype = c_int
libssnode.ssn_io_eval_f32.argtypes = [c_void_p, c_void_p, c_long, _pp, c_void_p]
libssnode.ssn_io_eval_f64.argtypes = [c_void_p, c_void_p, c_long, _pp, c_void_p]
for _name in ('ssn_build_w_f32', 'ssn_build_w_f64', 'ssn_stimulus_f32', 'ssn_stimulus_f64',
              'ssn_io_eval_f32', 'ssn_io_eval_f64'):
    getattr(libssnode, _name).restype = c_int
for _name in ('ssn_probe_scatter_f32', 'ssn_probe_scatter_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    getattr(libssnode, _name).restype = c_int
libssnode.ssn_segment_sqnorms2_f32.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]
libssnode.ssn_segment_sqnorms2_f32.restype = c_int
libssnode.ssn_segment_sqnorms_f32.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p]        # (older form: no scratch)
libssnode.ssn_segment_sqnorms_f32.restype = c_int
libssnode.ssn_segment_sqnorms_ws_doubles.argtypes = [c_int]
libssnode.ssn_segment_sqnorms_ws_doubles.restype = c_long
libssnode.ssn_interpolate_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]
libssnode.ssn_interpolate_f32.restype = c_int
for _name in ('ssn_philox_amp_f32', 'ssn_philox_amp_f64'):
    getattr(libssnode, _name).argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p, c_void_p, c_void_p,
                                          ctypes.c_ulonglong, c_int, c_int, c_void_p]
    getattr(libssnode, _name).restype = c_int
for _name in ('ssn_penalty_means_f32', 'ssn_penalty_means_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p, c_void_p, c_long, c_double, c_double, c_void_p, c_void_p, c_void_p]
    getattr(libssnode, _name).restype = c_int
for _name in ('ssn_penalty_means_probe_f32', 'ssn_penalty_means_probe_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p, c_void_p, c_long, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    getattr(libssnode, _name).restype = c_int
for _name in ('ssn_lu_solve_f32', 'ssn_lu_solve_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    getattr(libssnode, _name).restype = c_int
for _name in ('ssn_weight_grad_f32', 'ssn_weight_grad_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_void_p]
    getattr(libssnode, _name).restype = c_int
libssnode.ssn_build_w_philox_f32.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, POINTER(c_float), POINTER(c_float),
                                             POINTER(c_float), c_void_p, c_void_p, c_int, c_int, c_void_p]
libssnode.ssn_build_w_philox_f64.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, POINTER(c_double), POINTER(c_double),
                                             POINTER(c_double), c_void_p, c_void_p, c_int, c_int, c_void_p]
libssnode.ssn_build_w_philox_f32.restype = libssnode.ssn_build_w_philox_f64.restype = c_int
class GenInputs(Structure):
    """``ssn_gen_inputs`` of include/ssnode_mi355x.h."""
    _fields_ = [
        ('seed', ctypes.c_ulonglong), ('off_z', ctypes.c_ulonglong), ('off_zin', ctypes.c_ulonglong),
        ('J', POINTER(c_float)), ('D', POINTER(c_float)), ('S', POINTER(c_float)),
        ('bw', c_void_p), ('con', c_void_p), ('smoothness', c_float), ('v', c_void_p), ('bernoulli', c_int),
        ('W', c_void_p), ('z', c_void_p), ('zin', c_void_p), ('amp', c_void_p), ('ext', c_void_p),
        ('B', c_int), ('NB', c_int), ('N', c_int),
    ]


libssnode.ssn_gen_inputs_philox_f32.argtypes = [POINTER(GenInputs), c_void_p]
libssnode.ssn_gen_inputs_philox_f32.restype = c_int
libssnode.ssn_weight_grad_scaled_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_void_p, c_float, c_void_p]
libssnode.ssn_weight_grad_scaled_f32.restype = c_int
for _name in ('ssn_philox_uniform_f32', 'ssn_philox_uniform_f64'):
    getattr(libssnode, _name).argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p, ctypes.c_ulonglong, c_void_p]
    getattr(libssnode, _name).restype = c_int

class GenParams(Structure):
    """``ssn_gen_params`` of include/ssnode_mi355x.h."""
    _fields_ = [
        ('io_type', c_int), ('seqlen', c_int), ('skip_steps', c_int), ('kernel', c_int),
        ('k', c_double), ('n', c_double),
        ('tau_E', c_double), ('tau_I', c_double), ('dt', c_double),
        ('rate_soft_bound', c_double), ('rate_hard_bound', c_double),
        ('rate_penalty_threshold', c_double),
    ]


_gp = POINTER(GenParams)
libssnode.ssn_gen_supported.argtypes = [c_int, c_int]
libssnode.ssn_gen_supported.restype = c_int
libssnode.ssn_gen_forward_variant.argtypes = [c_int, c_int, c_int, c_int, c_int, c_void_p]
libssnode.ssn_gen_forward_variant.restype = c_int
for _name in ('ssn_gen_forward_f32', 'ssn_gen_forward_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p] * 7 + [c_int, c_int, c_int, _gp, c_void_p]
    getattr(libssnode, _name).restype = c_int
for _name in ('ssn_gen_backward_f32', 'ssn_gen_backward_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p] * 4 + [c_double, c_double, c_int, c_int, c_int, _gp, c_void_p]
    getattr(libssnode, _name).restype = c_int
for _name in ('ssn_gen_backward_ext_f32', 'ssn_gen_backward_ext_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p] * 5 + [c_double, c_double, c_int, c_int, c_int, _gp, c_void_p]
    getattr(libssnode, _name).restype = c_int
libssnode.ssn_gen_backward_max_f32.argtypes = [c_void_p] * 6 + [POINTER(c_int), c_double, c_double, c_int, c_int, c_int, _gp, c_void_p]
libssnode.ssn_gen_backward_max_f32.restype = c_int
libssnode.ssn_gen_backward_fused_supported.argtypes = [c_int, c_int, c_int, _gp, c_float]
libssnode.ssn_gen_backward_fused_supported.restype = c_int
libssnode.ssn_gen_backward_fused_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                                 c_double, c_double, c_int, c_int, c_int, _gp, c_void_p]
libssnode.ssn_gen_backward_fused_f32.restype = c_int
libssnode.ssn_jds_grad_f32.argtypes = [c_void_p, c_void_p, POINTER(c_float), POINTER(c_float), POINTER(c_float),
                                       c_void_p, c_int, c_int, c_void_p]
libssnode.ssn_jds_grad_f64.argtypes = [c_void_p, c_void_p, POINTER(c_double), POINTER(c_double), POINTER(c_double),
                                       c_void_p, c_int, c_int, c_void_p]
libssnode.ssn_jds_grad_f32.restype = c_int
libssnode.ssn_jds_grad_f64.restype = c_int

class OptParams(Structure):
    """``ssn_opt_params`` of include/ssnode_mi355x.h."""
    _fields_ = [
        ('kind', c_int), ('step', c_int), ('clip', c_int), ('reserved', c_int),
        ('learning_rate', c_double), ('beta1', c_double), ('beta2', c_double), ('epsilon', c_double),
        ('rho', c_double),
        ('reg_l2_penalty', c_double), ('reg_l1_penalty', c_double),
        ('reg_l2_decay', c_double), ('reg_l1_decay', c_double),
        ('clip_lo', c_double), ('clip_hi', c_double),
    ]


_ip = POINTER(c_int)


class CriticStep(Structure):
    """``ssn_critic_step`` of include/ssnode_mi355x.h."""
    _fields_ = [
        ('params', c_void_p), ('dims', _ip), ('layer_norm', _ip), ('nlayers', c_int), ('leak', c_float),
        ('xg', c_void_p), ('xd', c_void_p), ('cond', c_void_p), ('eps', c_void_p),
        ('n', c_int), ('hide_cell_type', c_int), ('precision', c_int), ('lmd', c_float),
        ('xp', c_void_p), ('grads', c_void_p), ('stats', c_void_p), ('dvals', c_void_p), ('workspace', c_void_p),
        ('opt_s1', c_void_p), ('opt_s2', c_void_p), ('opt', POINTER(OptParams)),
        ('seg_bounds', c_void_p), ('nseg', c_int), ('seg_ws', c_void_p),
        ('pens64', c_void_p),
        ('acc_dvals', c_void_p), ('tail', c_void_p),
    ]


libssnode.ssn_critic_step_run.argtypes = [POINTER(CriticStep), c_void_p]
libssnode.ssn_critic_step_run.restype = c_int
libssnode.ssn_critic_step_gated_run.argtypes = [POINTER(CriticStep), c_double, c_void_p]
libssnode.ssn_critic_step_gated_run.restype = c_int


class GenGrads(Structure):
    """``ssn_gen_grads`` of include/ssnode_mi355x.h."""
    _fields_ = [
        ('jds_part', c_void_p), ('B', c_int), ('nv', c_int),
        ('g_ext', c_void_p), ('ext_base', c_void_p), ('zin', c_void_p), ('NB', c_int), ('M', c_int),
        ('dmean', c_void_p), ('pens64', c_void_p), ('dynamics_cost', c_double), ('rate_cost', c_double),
        ('ws', c_void_p), ('out', c_void_p),
    ]


libssnode.ssn_gen_grads_ws_doubles.argtypes = []
libssnode.ssn_gen_grads_ws_doubles.restype = c_long
libssnode.ssn_gen_grads_f32.argtypes = [POINTER(GenGrads), c_void_p]
libssnode.ssn_gen_grads_f32.restype = c_int
libssnode.ssn_gen_apply_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, POINTER(OptParams), c_void_p, c_void_p,
                                        c_void_p, c_void_p]
libssnode.ssn_gen_apply_f32.restype = c_int
libssnode.ssn_critic_num_params.argtypes = [_ip, c_int]
libssnode.ssn_critic_num_params.restype = c_long
libssnode.ssn_critic_workspace_floats.argtypes = [_ip, c_int, c_int, c_int]
libssnode.ssn_critic_workspace_floats.restype = ctypes.c_size_t
libssnode.ssn_critic_forward.argtypes = [c_void_p, _ip, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                         c_int, c_void_p]
libssnode.ssn_critic_loss_grad.argtypes = [c_void_p, _ip, c_int] + [c_void_p] * 6 + [c_int, c_int, c_int, c_float,
                                           c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
libssnode.ssn_critic_input_grad.argtypes = [c_void_p, _ip, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                                            c_void_p, c_void_p, c_int, c_void_p]
libssnode.ssn_critic_forward_leaky.argtypes = [c_void_p, _ip, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                                               c_void_p, c_int, c_void_p]
libssnode.ssn_critic_loss_grad_leaky.argtypes = [c_void_p, _ip, c_int] + [c_void_p] * 6 + [
    c_int, c_int, c_int, c_float, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
libssnode.ssn_critic_input_grad_leaky.argtypes = [c_void_p, _ip, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_float,
                                                  c_void_p, c_void_p, c_void_p, c_int, c_void_p]
for _name in ('ssn_critic_forward_leaky', 'ssn_critic_loss_grad_leaky', 'ssn_critic_input_grad_leaky'):
    getattr(libssnode, _name).restype = c_int
libssnode.ssn_critic_accuracy.argtypes = [c_void_p, _ip, _ip, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                          c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
libssnode.ssn_critic_accuracy.restype = c_int
libssnode.ssn_critic_norm_workspace_floats.argtypes = [_ip, c_int, c_int, c_int]
libssnode.ssn_critic_norm_workspace_floats.restype = ctypes.c_size_t
libssnode.ssn_critic_forward_norm.argtypes = [c_void_p, _ip, _ip, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                              c_void_p, c_int, c_void_p]
libssnode.ssn_critic_loss_grad_norm.argtypes = [c_void_p, _ip, _ip, c_int] + [c_void_p] * 6 + [
    c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
libssnode.ssn_critic_input_grad_norm.argtypes = [c_void_p, _ip, _ip, c_int, c_void_p, c_void_p, c_int, c_int, c_float,
                                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p]
for _name in ('ssn_critic_forward_norm', 'ssn_critic_loss_grad_norm', 'ssn_critic_input_grad_norm'):
    getattr(libssnode, _name).restype = c_int
libssnode.ssn_optimizer_step.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_long, POINTER(OptParams), c_void_p]
for _name in ('ssn_critic_forward', 'ssn_critic_loss_grad', 'ssn_critic_input_grad', 'ssn_optimizer_step'):
    getattr(libssnode, _name).restype = c_int

class FFParams(Structure):
    """``ssn_ff_params`` of include/ssnode_mi355x.h."""
    _fields_ = [('nsam', c_int), ('nhid', c_int), ('ni', c_int), ('box', c_int),
                ('RF_l', c_double), ('RF_d', c_double), ('TH', c_double), ('TH_d', c_double),
                ('J', c_double), ('a', c_double)]


libssnode.ssn_ff_forward_f32.argtypes = [c_void_p] * 8 + [POINTER(FFParams), c_void_p]
libssnode.ssn_ff_forward_f32.restype = c_int
libssnode.ssn_ff_forward_sparse_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 5 + [POINTER(FFParams), c_void_p]
libssnode.ssn_ff_forward_sparse_f32.restype = c_int
libssnode.ssn_ff_backward_sparse_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 5 + [POINTER(FFParams), c_void_p]
libssnode.ssn_ff_backward_sparse_f32.restype = c_int
libssnode.ssn_ff_backward_f32.argtypes = [c_void_p] * 8 + [POINTER(FFParams), c_void_p]
libssnode.ssn_ff_backward_f32.restype = c_int
for _name, _ct in (('ssn_build_dw_f32', c_float), ('ssn_build_dw_f64', c_double)):
    getattr(libssnode, _name).argtypes = [c_void_p, POINTER(_ct), POINTER(_ct), POINTER(_ct), c_int, c_void_p, c_int,
                                           c_int, c_void_p]
    getattr(libssnode, _name).restype = c_int
for _name in ('ssn_ss_grad_system_f32', 'ssn_ss_grad_system_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                           POINTER(SolverParams), c_void_p, c_void_p, c_void_p]
    getattr(libssnode, _name).restype = c_int
libssnode.ssn_moment_sums_f32.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p]
libssnode.ssn_moment_sums_f32.restype = c_int
libssnode.ssn_moment_loss_grad_f32.argtypes = [c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_int, c_int,
                                               c_void_p, c_void_p, c_void_p]
libssnode.ssn_moment_loss_grad_f32.restype = c_int

#: every symbol include/ssnode_mi355x.h declares (checked by tests/test_abi.py)
DECLARED_SYMBOLS = (
    'solve_dynamics_asym_power_euler', 'solve_dynamics_asym_linear_euler', 'solve_dynamics_asym_tanh_euler',
    'io_pow', 'io_alin', 'io_atanh', 'rate_to_volt', 'dot',
    'ssn_abi_version', 'ssn_device_count', 'ssn_last_error', 'ssn_solver_fast_path',
    'ssn_solve_batch_f32', 'ssn_solve_batch_f64', 'ssn_solve_batch_f32_variant', 'ssn_solve_batch_f64_variant',
    'ssn_solve_batch_host_f32', 'ssn_solve_batch_host_f64',
    'ssn_build_w_f32', 'ssn_build_w_f64', 'ssn_stimulus_f32', 'ssn_stimulus_f64',
    'ssn_io_eval_f32', 'ssn_io_eval_f64',
    'ssn_gen_supported', 'ssn_gen_forward_variant', 'ssn_gen_forward_f32', 'ssn_gen_forward_f64',
    'ssn_gen_backward_f32', 'ssn_gen_backward_f64', 'ssn_jds_grad_f32', 'ssn_jds_grad_f64',
    'ssn_critic_num_params', 'ssn_critic_workspace_floats', 'ssn_critic_forward', 'ssn_critic_loss_grad',
    'ssn_critic_input_grad', 'ssn_optimizer_step',
    'ssn_ff_forward_sparse_f32', 'ssn_ff_backward_sparse_f32', 'ssn_ff_forward_f32', 'ssn_ff_backward_f32', 'ssn_moment_sums_f32', 'ssn_moment_loss_grad_f32',
    'ssn_build_dw_f32', 'ssn_build_dw_f64', 'ssn_ss_grad_system_f32', 'ssn_ss_grad_system_f64',
    'ssn_stimulus_amp_f32', 'ssn_stimulus_amp_f64', 'ssn_stimulus_hetero_f32', 'ssn_gen_backward_ext_f32', 'ssn_gen_backward_ext_f64',
    'ssn_critic_norm_workspace_floats', 'ssn_critic_forward_norm', 'ssn_critic_loss_grad_norm',
    'ssn_critic_input_grad_norm', 'ssn_philox_uniform_f32', 'ssn_philox_uniform_f64',
    'ssn_weight_grad_f32', 'ssn_weight_grad_f64', 'ssn_lu_solve_f32', 'ssn_lu_solve_f64',
    'ssn_penalty_means_f32', 'ssn_penalty_means_f64', 'ssn_penalty_means_probe_f32', 'ssn_penalty_means_probe_f64', 'ssn_philox_amp_f32', 'ssn_philox_amp_f64',
    'ssn_segment_sqnorms_f32', 'ssn_segment_sqnorms2_f32', 'ssn_interpolate_f32', 'ssn_probe_scatter_f32', 'ssn_probe_scatter_f64',
    'ssn_set_operand_precision', 'ssn_get_operand_precision', 'ssn_solve_batch_variant_for', 'ssn_segment_sqnorms_ws_doubles',
    'ssn_gen_backward_max_f32', 'ssn_gen_backward_fused_supported', 'ssn_gen_backward_fused_f32', 'ssn_weight_grad_scaled_f32', 'ssn_build_w_philox_f32', 'ssn_build_w_philox_f64',
    'ssn_critic_forward_leaky', 'ssn_critic_loss_grad_leaky', 'ssn_critic_input_grad_leaky', 'ssn_critic_accuracy', 'ssn_critic_step_run', 'ssn_critic_step_gated_run', 'ssn_gen_grads_ws_doubles', 'ssn_gen_grads_f32', 'ssn_gen_apply_f32',
    'ssn_gen_inputs_philox_f32',
    'ssn_mt19937_random_sample_f32', 'ssn_mt19937_random_sample_f64', 'ssn_mt19937_jump_poly',
    'ssn_mt19937_random_sample_begin_f32', 'ssn_mt19937_random_sample_begin_f64', 'ssn_mt19937_random_sample_finish',
    'ssn_mt19937_plan', 'ssn_build_w_mt19937_begin_f32', 'ssn_build_w_devparams_f32', 'ssn_mt19937_random_sample_tail_begin_f32',
    'ssn_build_w_mt19937_tail_begin_f32', 'ssn_mt19937_plan_tail',
    'ssn_critic_num_params_act', 'ssn_critic_forward_act', 'ssn_critic_loss_grad_act', 'ssn_critic_input_grad_act',
    'ssn_critic_accuracy_act',
)

libssnode.ssn_critic_num_params_act.argtypes = [c_void_p, c_void_p, c_int]
libssnode.ssn_critic_num_params_act.restype = c_long
libssnode.ssn_critic_forward_act.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                             c_void_p, c_void_p, c_int, c_void_p]
libssnode.ssn_critic_loss_grad_act.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int] * 3 + \
    [c_float, c_int] + [c_void_p] * 4 + [c_int, c_void_p]
libssnode.ssn_critic_input_grad_act.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                                c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
libssnode.ssn_critic_accuracy_act.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 4 + [c_int] * 3 + \
    [c_void_p] * 3 + [c_int, c_void_p]
for _name in ('ssn_critic_forward_act', 'ssn_critic_loss_grad_act', 'ssn_critic_input_grad_act', 'ssn_critic_accuracy_act'):
    getattr(libssnode, _name).restype = c_int

for _name in ('ssn_mt19937_random_sample_f32', 'ssn_mt19937_random_sample_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p, POINTER(c_int), ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong,
                                          c_void_p, c_void_p]
    getattr(libssnode, _name).restype = c_int
libssnode.ssn_mt19937_jump_poly.argtypes = [ctypes.c_ulonglong, c_void_p]
libssnode.ssn_mt19937_jump_poly.restype = c_int
for _name in ('ssn_mt19937_random_sample_begin_f32', 'ssn_mt19937_random_sample_begin_f64'):
    getattr(libssnode, _name).argtypes = [c_void_p, c_int, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong,
                                          c_void_p, c_void_p, POINTER(c_int)]
    getattr(libssnode, _name).restype = c_int
libssnode.ssn_build_w_mt19937_begin_f32.argtypes = [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float),
                                                     POINTER(c_float), c_void_p, c_void_p, c_int, c_void_p, POINTER(c_int)]
libssnode.ssn_build_w_mt19937_begin_f32.restype = c_int
libssnode.ssn_mt19937_random_sample_tail_begin_f32.argtypes = [c_void_p, c_int, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong,
                                                               c_void_p, c_int, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong,
                                                               c_void_p, c_void_p, POINTER(c_int)]
libssnode.ssn_mt19937_random_sample_tail_begin_f32.restype = c_int
libssnode.ssn_build_w_mt19937_tail_begin_f32.argtypes = [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float),
                                                         POINTER(c_float), c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                                         POINTER(c_int)]
libssnode.ssn_build_w_mt19937_tail_begin_f32.restype = c_int
libssnode.ssn_build_w_devparams_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]
libssnode.ssn_build_w_devparams_f32.restype = c_int
libssnode.ssn_mt19937_plan_tail.argtypes = [c_int, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong, c_int, ctypes.c_ulonglong,
                                            ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p]
libssnode.ssn_mt19937_plan_tail.restype = c_int
libssnode.ssn_mt19937_plan.argtypes = [c_int, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p]
libssnode.ssn_mt19937_plan.restype = c_int
libssnode.ssn_mt19937_random_sample_finish.argtypes = [c_int, c_void_p, POINTER(c_int)]
libssnode.ssn_mt19937_random_sample_finish.restype = c_int

libssnode.ssn_set_operand_precision.argtypes = [c_int]
libssnode.ssn_set_operand_precision.restype = c_int
libssnode.ssn_get_operand_precision.argtypes = []
libssnode.ssn_get_operand_precision.restype = c_int
libssnode.ssn_solve_batch_variant_for.argtypes = [c_int, c_int, c_int, c_int, POINTER(SolverParams)]
libssnode.ssn_solve_batch_variant_for.restype = c_int

#: names of the generator kernel families (`ssn_gen_params.kernel`, include/ssnode_mi355x.h) as the Python API and the
#: command line spell them: TuningCurveGenerator(gen_kernel=...), --gen-kernel
GEN_KERNELS = collections.OrderedDict([
    ('auto', 0),          # library's choice under the current operand-precision setting
    ('tile', 1),          # VALU tile kernels, fp32 operands
    ('mfma-fp32', 2),     # fp32 matrix-core kernels (two 4-stimulus groups per workgroup), fp32 operands
    ('mfma-fp32-1g', 3),  # the same, one group per workgroup (few draws)
    ('split-wide', 4),    # fp16-split matrix-core kernel, all 8 stimuli in one chain (W and state 22-23 bits)
    ('split-1g', 5),      # fp16-split, one group per workgroup (state exact)
    ('split-alt', 6),     # fp16-split, two alternating groups (state exact)
    ('duo', 8),           # fp16-split, two draws per workgroup (W and state 23 bits by round to nearest)
    ('duo-fused', 9),     # forward as 'duo'; backward = ssn_gen_backward_fused_f32 (adjoint sweep + dL/dW in one launch,
                          # one draw per workgroup); a host-side choice: ssn_gen_params.kernel carries 8
])
GEN_KERNEL_FUSED = 9
OPERAND_PRECISIONS = {'fp32': 0, 'split': 1}


def gen_kernel_code(kernel):
    """'auto' / 'duo' / ... or the integer code -> the integer code of `ssn_gen_params.kernel`."""
    if isinstance(kernel, str):
        if kernel not in GEN_KERNELS:
            raise ValueError('Unknown generator kernel {!r}; choose from {}'.format(kernel, ', '.join(GEN_KERNELS)))
        return GEN_KERNELS[kernel]
    kernel = int(kernel)
    if kernel not in GEN_KERNELS.values():
        raise ValueError('Unknown generator kernel code {}'.format(kernel))
    return kernel


def set_operand_precision(mode):
    """'fp32' or 'split': operand precision of the library's AUTOMATIC kernel choice (`ssn_set_operand_precision`);
    returns the previous setting by name.  Explicit kernel choices are not affected."""
    prev = libssnode.ssn_set_operand_precision(OPERAND_PRECISIONS[mode])
    return 'split' if prev else 'fp32'


def get_operand_precision():
    return 'split' if libssnode.ssn_get_operand_precision() else 'fp32'


class GPUUnavailableError(RuntimeError):
    """No usable HIP device: the product path refuses to run (no CPU fallback)."""


class SSNLibraryError(RuntimeError):
    """A libssnode entry point returned SSN_ERR_BASE + hipError_t."""


def last_error():
    return libssnode.ssn_last_error().decode()


def stream_ptr():
    """The current torch stream of the current device as the `void *stream` of the additive ABI.  (Goes to the two
    C entry points directly: `torch.cuda.current_stream().cuda_stream` costs several microseconds of Python per call,
    and every launch of the GAN loop needs it.)"""
    import torch
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def check(rc, what):
    """Raise on a non-zero status of an additive-ABI call."""
    if rc != 0:
        raise SSNLibraryError('{} failed: status {} ({})'.format(what, rc, last_error()))


_DEVICES_SEEN = 0


def require_gpu():
    """Fail loudly when the HIP runtime sees no device (a positive count is remembered: the hot loop asks per launch)."""
    global _DEVICES_SEEN
    if _DEVICES_SEEN > 0:
        return _DEVICES_SEEN
    n = libssnode.ssn_device_count()
    if n <= 0:
        raise GPUUnavailableError(
            'libssnode found no HIP device (ssn_device_count() = {}; {}). '
            'tc_gan_amd has no CPU fallback.'.format(n, last_error()))
    _DEVICES_SEEN = n
    return n
