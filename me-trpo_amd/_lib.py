"""ctypes binding of libmetrpo.so (include/metrpo.h).  There is NO CPU fallback: if the HIP library
is missing or fails to load, importing this module raises."""
import ctypes as C
import os

import torch  # noqa: F401  -- must be imported first so that libmetrpo.so binds to torch's libamdhip64.so.7

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmetrpo.so")
MAX_LAYERS = 6

ENV_IDS = {'swimmer': 0, 'half_cheetah': 1, 'half-cheetah': 1, 'ant': 2, 'humanoid': 3, 'hopper': 4, 'snake': 5}
SAM_MODES = {'step_rand': 0, 'eps_rand': 1, 'model_mean_std': 2, 'model_mean': 3, 'model_med': 4, 'one_model': 5}
ACTS = {'identity': 0, 'tf.identity': 0, 'relu': 1, 'tf.nn.relu': 1, 'tanh': 2, 'tf.nn.tanh': 2, 'tf.tanh': 2}


class Dims(C.Structure):
    _fields_ = [('env', C.c_int32), ('ns', C.c_int32), ('na', C.c_int32), ('n_models', C.c_int32),
                ('dyn_n_hidden', C.c_int32), ('dyn_hidden', C.c_int32 * MAX_LAYERS), ('dyn_act', C.c_int32 * MAX_LAYERS),
                ('n_drop', C.c_int32), ('pol_n_hidden', C.c_int32), ('pol_hidden', C.c_int32 * MAX_LAYERS)]


class RolloutArgs(C.Structure):
    _fields_ = [('B', C.c_int32), ('T', C.c_int32), ('H', C.c_int32), ('sam_mode', C.c_int32), ('determ', C.c_int32),
                ('eval_all_heads', C.c_int32), ('d_pool', C.c_void_p), ('n_pool', C.c_int32), ('seed', C.c_uint64),
                ('stream_offset', C.c_uint64), ('d_eps', C.c_void_p), ('d_model_idx', C.c_void_p),
                ('d_sel_noise', C.c_void_p), ('d_reset_idx', C.c_void_p), ('d_reset_model', C.c_void_p),
                ('d_obs', C.c_void_p), ('d_act', C.c_void_p), ('d_rew', C.c_void_p), ('d_mean', C.c_void_p),
                ('d_done', C.c_void_p), ('d_tpath', C.c_void_p), ('d_last_obs', C.c_void_p),
                # continuation (ABI 2)
                ('t0', C.c_int32), ('d_init_obs', C.c_void_p), ('d_init_ts', C.c_void_p), ('d_init_model', C.c_void_p),
                ('d_last_ts', C.c_void_p), ('d_last_model', C.c_void_p), ('d_stop', C.c_void_p),
                # in-launch stop rule (ABI 4)
                ('stop_batch', C.c_int64), ('d_stop_cum', C.c_void_p)]


class Batch(C.Structure):
    _fields_ = [('d_obs', C.c_void_p), ('d_act', C.c_void_p), ('d_adv', C.c_void_p), ('d_old_mean', C.c_void_p),
                ('d_old_log_std', C.c_void_p), ('old_log_std_stride', C.c_int32), ('d_valid', C.c_void_p),
                ('N', C.c_int64), ('inv_n_global', C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class TrpoParams(C.Structure):
    _fields_ = [('max_kl', C.c_double), ('cg_iters', C.c_int32), ('reg_coeff', C.c_double),
                ('backtrack_ratio', C.c_double), ('max_backtracks', C.c_int32), ('accept_violation', C.c_int32),
                ('residual_tol', C.c_double), ('allreduce', ALLREDUCE_FN), ('allreduce_user', C.c_void_p),
                ('explicit_final_hvp', C.c_int32)]


class TrainParams(C.Structure):
    _fields_ = [('lr', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double), ('eps', C.c_double),
                ('reg_constant', C.c_double), ('batch_size', C.c_int32)]


class TrpoDiag(C.Structure):
    _fields_ = [('loss_before', C.c_double), ('loss', C.c_double), ('kl', C.c_double), ('beta', C.c_double),
                ('n_backtrack', C.c_int32), ('accepted', C.c_int32), ('cg_iters_run', C.c_int32)]


# every symbol include/metrpo.h declares: name -> (restype, argtypes)
_P, _I, _L, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SYMBOLS = {
    'metrpo_abi_version': (_I, []),
    'metrpo_status_string': (C.c_char_p, [_I]),
    'metrpo_create': (_I, [C.POINTER(_P), _I, C.POINTER(Dims)]),
    'metrpo_destroy': (_I, [_P]),
    'metrpo_last_error': (C.c_char_p, [_P]),
    'metrpo_dyn_param_count': (_I, [_P]),
    'metrpo_policy_param_count': (_I, [_P]),
    'metrpo_set_dynamics': (_I, [_P, _P, _P, _P, _P, _P, _P]),
    'metrpo_set_policy': (_I, [_P, _P, _P]),
    'metrpo_get_policy': (_I, [_P, _P, _P]),
    'metrpo_policy_actions': (_I, [_P, _P, _P, _I, _P, _P, _P]),
    'metrpo_step': (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    'metrpo_rollout': (_I, [_P, C.POINTER(RolloutArgs), _P]),
    'metrpo_comm_get_unique_id': (_I, [_P]),
    'metrpo_comm_init': (_I, [_P, _P, _I, _I]),
    'metrpo_comm_destroy': (_I, [_P]),
    'metrpo_allreduce_sum_f64': (_I, [_P, _P, _L, _P]),
    'metrpo_comm_ipc_export': (_I, [_P, _P]),
    'metrpo_comm_ipc_attach': (_I, [_P, _P, _I, _I]),
    'metrpo_comm_ipc_detach': (_I, [_P]),
    'metrpo_comm_set_timeout_ms': (_I, [_P, _L]),
    'metrpo_comm_transport': (_I, [_P]),
    'metrpo_comm_check': (_I, [_P, _P]),
    'metrpo_sampler_progress': (_I, [_P, _P, _P, _I, _I, _I, _L, _P, _P, _P, _P]),
    'metrpo_validation_cost': (_I, [_P, _P, _I, _I, _D, _P, _P]),
    'metrpo_gae': (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _D, _D, _P, _P, _P, _P, _P]),
    'metrpo_process_begin': (_I, [_P, _P, _P, _L, _P]),
    'metrpo_center_advantages': (_I, [_P, _P, _P, _L, _P, _P]),
    'metrpo_baseline_gram': (_I, [_P, _P, _P, _P, _P, _L, _P, _P, _P]),
    'metrpo_baseline_solve': (_I, [_P, _P, _P, _D, _P, _P]),
    'metrpo_loss_grad': (_I, [_P, C.POINTER(Batch), _P, _P]),
    'metrpo_fvp': (_I, [_P, C.POINTER(Batch), _P, _P, _P]),
    'metrpo_loss_kl': (_I, [_P, C.POINTER(Batch), _P, _P, _P]),
    'metrpo_trpo_update': (_I, [_P, C.POINTER(Batch), C.POINTER(TrpoParams), C.POINTER(TrpoDiag), _P, _P, _P]),
    'metrpo_trpo_update_begin': (_I, [_P, C.POINTER(Batch), C.POINTER(TrpoParams), _I, _P, _P, _P]),
    'metrpo_trpo_update_end': (_I, [_P, C.POINTER(TrpoDiag), C.POINTER(C.c_int32), _P]),
    'metrpo_dyn_train_reset': (_I, [_P, _P]),
    'metrpo_dyn_train_step': (_I, [_P, _P, _P, C.POINTER(TrainParams), _P, _P]),
    'metrpo_dyn_eval_losses': (_I, [_P, _P, _P, _L, _D, _P, _P]),
    'metrpo_get_dynamics': (_I, [_P, _P, _P]),
    'metrpo_set_dynamics_model': (_I, [_P, _I, _P, _P]),
    'metrpo_set_normalizers': (_I, [_P, _P, _P, _P, _P, _P]),
    'metrpo_rms_accumulate': (_I, [_P, _P, _L, _I, _P, _P, _P]),
    'metrpo_bptt_grad': (_I, [_P, _P, _I, _I, _D, _P, _P, _P]),
    'metrpo_set_exclusive': (_I, [_P, _I]),
    'metrpo_set_option': (_I, [_P, C.c_char_p, C.c_char_p]),
    'metrpo_get_option': (_I, [_P, C.c_char_p, C.c_char_p, _I]),
    'metrpo_option_name': (C.c_char_p, [_I]),
    'metrpo_last_rollout_kernel': (_I, [_P]),
    'metrpo_rollout_note': (C.c_char_p, [_P]),
    'metrpo_policy_adam_reset': (_I, [_P, _P]),
    'metrpo_policy_adam_step': (_I, [_P, _P, _D, _D, _D, _D, _D, _P]),
}
# diagnostics hooks exported besides the header's ABI (used by tests to cross-check the two rollout kernels)
EXTRA_SYMBOLS = {
    'metrpo_rollout_generic': (_I, [_P, C.POINTER(RolloutArgs), _P]),
    'metrpo_has_mfma_path': (_I, [_P]),
    'metrpo_set_update_path': (_I, [_P, _I]),
    'metrpo_update_path': (_I, [_P, _L]),
    'metrpo_set_rollout_variant': (_I, [_P, _I]),
    'metrpo_set_det_path': (_I, [_P, _I]),
    'metrpo_probe_peaks': (_I, [_P, _P, _P]),
    'metrpo_schedulable_cus': (_I, [_P, _P]),
    'metrpo_debug_fvp_us': (_I, [_P, _P, _P]),
    'metrpo_debug_persist_stats': (_I, [_P, _P, _I, _P]),
    'metrpo_debug_ws_retired': (_I, [_P, _P, _I]),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmetrpo.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for table in (SYMBOLS, EXTRA_SYMBOLS):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
    if lib.metrpo_abi_version() != 4:
        raise ImportError("libmetrpo.so ABI version mismatch")
    return lib


lib = load()


class MetrpoError(RuntimeError):
    pass


def check(rc, ctx=None):
    if rc != 0:
        msg = lib.metrpo_status_string(rc).decode()
        if ctx:
            msg += ": " + lib.metrpo_last_error(ctx).decode()
        raise MetrpoError(msg)
