"""Thin tensor-level wrapper of one metrpo_ctx (one GPU).  torch is used for device memory and
streams only; all arithmetic happens in libmetrpo.so.  Every method enqueues on torch's current
stream of the engine's device."""
import ctypes as C
from collections import namedtuple

import numpy as np
import torch

from . import _lib
from ._lib import lib, check

METRPO_UNSET = -100            # include/metrpo.h: metrpo_get_option of a known key without a value

Trajectory = namedtuple('Trajectory', 'obs act rew mean done tpath last_obs B T H')
"""Time-major device tensors of one rollout: obs [T,B,ns], act [T,B,na] (unclipped), rew [T,B],
mean [T,B,na], done [T,B] uint8, tpath [T,B] int32, last_obs [B,ns]."""


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32(t, dev, shape=None):
    if t is None:
        return None
    t = torch.as_tensor(t, device=dev).to(torch.float32).contiguous()
    if shape is not None:
        assert tuple(t.shape) == tuple(shape), "expected shape %s, got %s" % (shape, tuple(t.shape))
    return t


def _i32(t, dev, shape=None):
    if t is None:
        return None
    t = torch.as_tensor(t, device=dev).to(torch.int32).contiguous()
    if shape is not None:
        assert tuple(t.shape) == tuple(shape), "expected shape %s, got %s" % (shape, tuple(t.shape))
    return t


class Engine(object):
    """Owns the metrpo_ctx for one (env, ensemble shape, policy shape) on one GPU."""

    def __init__(self, env, n_models, dyn_hidden, pol_hidden, ns=None, na=None, n_drop=None, dyn_act='relu',
                 device=None, prediction_type='state_change', use_logit_weights=False):
        # the dynamics-model variants the reference defines but none of its six-env params files selects have no kernel here: say so by name
        # instead of silently evaluating the 'state_change' arithmetic (dynamics_model.prediction_type / .use_logit_weights of params-*.json)
        if prediction_type != 'state_change':
            raise ValueError("dynamics_model.prediction_type=%r is not implemented: only 'state_change' (training.py:257: s' = diff_mean + diff_std * out + s); "
                             "'second_derivative' (training.py:259-264) and the '*_goal' variants (training.py:265-268) have no kernel in this library"
                             % (prediction_type,))
        if use_logit_weights:
            raise ValueError("dynamics_model.use_logit_weights=True is not implemented (the sigmoid input gate of training.py:234-242, 212-213); "
                             "every shipped params file of the six envs sets it to false")
        if not torch.cuda.is_available():
            raise RuntimeError("metrpo_amd needs an AMD GPU (gfx950); there is no CPU fallback")
        self.env_name = env.replace('-', '_')
        specs = {'swimmer': (10, 2, 2), 'half_cheetah': (18, 6, 1), 'ant': (29, 8, 2), 'humanoid': (55, 21, 0),
                 'hopper': (11, 3, 0), 'snake': (14, 4, 2)}
        d_ns, d_na, d_drop = specs[self.env_name]
        self.ns, self.na = ns or d_ns, na or d_na
        self.n_drop = d_drop if n_drop is None else n_drop
        self.K = int(n_models)
        self.dyn_hidden, self.pol_hidden = list(dyn_hidden), list(pol_hidden)
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        acts = [dyn_act] * len(self.dyn_hidden) if isinstance(dyn_act, str) else list(dyn_act)
        d = _lib.Dims()
        d.env, d.ns, d.na, d.n_models = _lib.ENV_IDS[self.env_name], self.ns, self.na, self.K
        d.dyn_n_hidden, d.n_drop, d.pol_n_hidden = len(self.dyn_hidden), self.n_drop, len(self.pol_hidden)
        for i, h in enumerate(self.dyn_hidden):
            d.dyn_hidden[i], d.dyn_act[i] = h, _lib.ACTS[acts[i]]
        for i, h in enumerate(self.pol_hidden):
            d.pol_hidden[i] = h
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = lib.metrpo_create(C.byref(self._ctx), self.device.index, C.byref(d))
        check(rc, self._ctx if self._ctx else None)
        self.dyn_param_count = lib.metrpo_dyn_param_count(self._ctx)
        self.P = lib.metrpo_policy_param_count(self._ctx)
        self.pol_dims = [self.ns] + self.pol_hidden + [self.na]
        self.has_mfma_path = bool(lib.metrpo_has_mfma_path(self._ctx))
        self._cb_keepalive = None
        self.comm_world = 0                  # > 0 once comm_init attached an RCCL communicator

    def __del__(self):
        ctx = getattr(self, '_ctx', None)
        if ctx:
            lib.metrpo_destroy(ctx)
            self._ctx = None

    def set_rollout_variant(self, v):
        """Test hook: 0 = fastest rollout kernel available, 1 = head-per-wave MFMA kernel (for large nets: the step-wise
        GEMM path even where the resident kernel applies), 2 = cooperative kernel in its
        two-workgroups-per-CU instantiation at any batch size.  Returns the variant that
        will run: 3 step-wise GEMM (large nets), 2 cooperative-heads MFMA, 1 head-per-wave MFMA, 0 generic."""
        self._variant = int(v)
        return int(lib.metrpo_set_rollout_variant(self._ctx, int(v)))

    def set_update_path(self, use_mfma):
        """Test hook: False forces the generic (VALU) policy-update kernels, True selects the fastest path of the shape (fused
        MFMA kernels, else the GEMM path for large N), 'gemm' forces the GEMM path (policy_gemm.hip).  Returns True if the fused
        MFMA kernels are active, 'gemm' if the GEMM path is forced, else False."""
        code = 2 if use_mfma == 'gemm' else int(bool(use_mfma))
        r = int(lib.metrpo_set_update_path(self._ctx, code))
        return 'gemm' if r == 2 else bool(r)

    # ------------------------------------------------------------------ multi-GPU: RCCL communicator owned by the ctx
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes (ncclGetUniqueId) that rank 0 ships to the other ranks before every rank calls comm_init."""
        buf = C.create_string_buffer(128)
        check(lib.metrpo_comm_get_unique_id(buf), None)
        return buf.raw

    def comm_init(self, unique_id, world, rank):
        """Collective: attach an RCCL communicator; metrpo_trpo_update then issues its all-reduces itself (no host callback)."""
        assert len(unique_id) == 128
        with torch.cuda.device(self.device):
            self._chk(lib.metrpo_comm_init(self._ctx, C.c_char_p(unique_id), int(world), int(rank)))
        self.comm_world = int(world)

    def comm_destroy(self):
        self._chk(lib.metrpo_comm_destroy(self._ctx))
        self.comm_world = 0

    # ---- one-shot direct all-reduce over peer-mapped receive regions (comm.hip / xchg_device.h)
    IPC_BLOB_BYTES = 128

    def comm_ipc_export(self):
        """Allocate + zero this rank's receive region; -> 128 opaque bytes (IPC handle) to all-gather over the ranks."""
        buf = C.create_string_buffer(self.IPC_BLOB_BYTES)
        with torch.cuda.device(self.device):
            self._chk(lib.metrpo_comm_ipc_export(self._ctx, buf))
        return buf.raw

    def comm_ipc_attach(self, blobs, world, rank):
        """blobs: the ranks' export blobs concatenated in rank order.  Maps the peers' regions; from then on the sum all-reduces of the
        path are one-shot exchanges issued by libmetrpo.so (inside metrpo_trpo_update: in the tail of the reduction kernels)."""
        assert len(blobs) == self.IPC_BLOB_BYTES * int(world)
        with torch.cuda.device(self.device):
            self._chk(lib.metrpo_comm_ipc_attach(self._ctx, C.c_char_p(blobs), int(world), int(rank)))
        self.comm_world = int(world)

    def comm_ipc_detach(self):
        self._chk(lib.metrpo_comm_ipc_detach(self._ctx))
        if not lib.metrpo_comm_transport(self._ctx):
            self.comm_world = 0

    def comm_set_timeout_ms(self, ms):
        self._chk(lib.metrpo_comm_set_timeout_ms(self._ctx, int(ms)))

    def comm_transport(self):
        return {0: None, 1: 'rccl', 2: 'one-shot'}[int(lib.metrpo_comm_transport(self._ctx))]

    def comm_check(self):
        """Synchronise and raise if an exchange timed out (a rank that never arrived)."""
        self._chk(lib.metrpo_comm_check(self._ctx, self._stream()))

    def allreduce_sum_(self, t):
        """In-place sum of a float64 device tensor over the ranks of the attached communicator (stream-ordered)."""
        assert t.dtype == torch.float64 and t.is_cuda and t.is_contiguous()
        self._chk(lib.metrpo_allreduce_sum_f64(self._ctx, _ptr(t), t.numel(), self._stream()))
        return t

    def set_exclusive(self, exclusive):
        """False: other compute processes share this GPU (several ranks per device) -- the kernels whose workgroups wait on each other inside one
        launch (resident rollout / validation, migrating tiles) are never selected.  Default True."""
        self._chk(lib.metrpo_set_exclusive(self._ctx, int(bool(exclusive))))

    def set_option(self, key, value=None):
        """Variant / tuning switch of THIS engine (`metrpo_set_option`): `key` is one of `Engine.option_names()` (the former METRPO_<KEY> environment
        variables; the environment only fills the defaults when the engine is created), `value` a string / number, None unsets.  Read at the next launch."""
        self._chk(lib.metrpo_set_option(self._ctx, str(key).encode(), None if value is None else str(value).encode()))

    def get_option(self, key):
        """Current value of a switch as a string, None when unset."""
        buf = C.create_string_buffer(256)
        n = lib.metrpo_get_option(self._ctx, str(key).encode(), buf, 256)
        if n == METRPO_UNSET:                     # a known key without a value; an unknown key is METRPO_EINVAL and raises
            return None
        if n < 0:
            self._chk(n)
        return buf.value.decode()

    @staticmethod
    def option_names():
        out, i = [], 0
        while True:
            nm = lib.metrpo_option_name(i)
            if nm is None:
                return out
            out.append(nm.decode()); i += 1

    def schedulable_cus(self):
        """CUs that actually run this process's waves (a census kernel; CU masks and partitions count), measured once per engine."""
        return int(lib.metrpo_schedulable_cus(self._ctx, self._stream()))

    def probe_peaks(self):
        """Measured (f32 MFMA TFLOP/s, HBM copy GB/s) of this device: register-resident MFMA issue loop, 1 GiB streaming copy."""
        out = (C.c_double * 2)()
        self._chk(lib.metrpo_probe_peaks(self._ctx, out, self._stream()))
        return float(out[0]), float(out[1])

    def fvp_kernel_us(self):
        """Diagnostics: mean duration (us) and count of the Fisher-vector-product KERNEL launches recorded since the last call while option
        TIME_FVP was set (HIP events around the kernel itself inside the update's launch sequence)."""
        us, n = C.c_double(0.0), C.c_int32(0)
        self._chk(lib.metrpo_debug_fvp_us(self._ctx, C.byref(us), C.byref(n)))
        return float(us.value), int(n.value)

    def retired_workspaces(self, sweep=False):
        """Diagnostics: (count, bytes) of outgrown workspaces the context keeps until destroy (a launch entry point never frees: hipFree waits for every stream);
        sweep=True frees them now (synchronising)."""
        b = C.c_ulonglong(0)
        n = int(lib.metrpo_debug_ws_retired(self._ctx, C.byref(b), int(bool(sweep))))
        return n, int(b.value)

    def update_path(self, N):
        """Kernel family the policy update of an N-sample batch runs on: 'mfma' (fused), 'gemm' or 'generic'."""
        return {1: 'mfma', 2: 'gemm', 0: 'generic'}[int(lib.metrpo_update_path(self._ctx, int(N)))]

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, rc):
        check(rc, self._ctx)

    # ------------------------------------------------------------------ weights
    def set_dynamics(self, params, in_mean, in_std, diff_mean, diff_std):
        """params [K, dyn_param_count] (per model: W0,b0,...,Wout,bout, W row-major (n_in,n_out))."""
        dev = self.device
        p = _f32(params, dev, (self.K, self.dyn_param_count))
        a = _f32(in_mean, dev, (self.ns + self.na,)); b = _f32(in_std, dev, (self.ns + self.na,))
        c = _f32(diff_mean, dev, (self.ns,)); e = _f32(diff_std, dev, (self.ns,))
        self._chk(lib.metrpo_set_dynamics(self._ctx, _ptr(p), _ptr(a), _ptr(b), _ptr(c), _ptr(e), self._stream()))
        torch.cuda.current_stream(dev).synchronize()     # inputs may be temporaries

    def set_dynamics_layers(self, Ws, bs, in_mean, in_std, diff_mean, diff_std):
        """Ws[l]: [K, n_in, n_out], bs[l]: [K, n_out] -> packs the flat per-model layout."""
        parts = []
        for W, b in zip(Ws, bs):
            W = torch.as_tensor(W); b = torch.as_tensor(b)
            parts += [W.reshape(self.K, -1), b.reshape(self.K, -1)]
        self.set_dynamics(torch.cat([p.to(torch.float32) for p in parts], dim=1), in_mean, in_std, diff_mean, diff_std)

    def set_policy(self, theta):
        self._close_open_update()
        t = _f32(theta, self.device, (self.P,))
        self._chk(lib.metrpo_set_policy(self._ctx, _ptr(t), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    def get_policy(self):
        self._close_open_update()
        out = torch.empty(self.P, dtype=torch.float32, device=self.device)
        self._chk(lib.metrpo_get_policy(self._ctx, _ptr(out), self._stream()))
        return out

    # ------------------------------------------------------------------ step-level API
    def policy_actions(self, obs, eps=None):
        dev = self.device
        obs = _f32(obs, dev); B = obs.shape[0]
        eps = _f32(eps, dev, (B, self.na))
        actions = torch.empty(B, self.na, dtype=torch.float32, device=dev)
        mean = torch.empty_like(actions)
        self._chk(lib.metrpo_policy_actions(self._ctx, _ptr(obs), _ptr(eps), B, _ptr(actions), _ptr(mean), self._stream()))
        return actions, mean

    def step(self, s, a, sam_mode, model_idx=None, noise=None, want_all=False):
        dev = self.device
        s = _f32(s, dev); B = s.shape[0]
        a = _f32(a, dev, (B, self.na))
        model_idx = _i32(model_idx, dev, (B,)); noise = _f32(noise, dev, (B, self.ns))
        s_next = torch.empty(B, self.ns, dtype=torch.float32, device=dev)
        rew = torch.empty(B, dtype=torch.float32, device=dev)
        done = torch.empty(B, dtype=torch.uint8, device=dev)
        nall = torch.empty(self.K, B, self.ns, dtype=torch.float32, device=dev) if want_all else None
        self._chk(lib.metrpo_step(self._ctx, _ptr(s), _ptr(a), B, _lib.SAM_MODES[sam_mode], _ptr(model_idx), _ptr(noise),
                                  _ptr(s_next), _ptr(rew), _ptr(done), _ptr(nall), self._stream()))
        return (s_next, rew, done, nall) if want_all else (s_next, rew, done)

    # ------------------------------------------------------------------ fused rollout
    def rollout(self, B, T, H, sam_mode, pool, determ=False, eval_all_heads=True, seed=0, stream_offset=0,
                eps=None, model_idx=None, sel_noise=None, reset_idx=None, reset_model=None, out=None,
                force_generic=False, t0=0, resume=None, last_state=None, stop=None, stop_batch=0, stop_cum=None):
        """`resume` = (obs [B,ns] f32, ts [B] i32, model [B] i32) continues a chunked rollout at global step `t0` (the supplied
        draw tensors are then indexed from this chunk's first step); `last_state` = (ts, model) int32 output tensors;
        `stop` = int32 device flag that turns the call into a no-op when set (metrpo_sampler_progress); `stop_batch` > 0 with `stop_cum` (float64 device
        scalar: samples completed in front of this call) lets a one-launch kernel family apply the sampler's stop rule inside the call (rows behind the stop
        step are then undefined)."""
        dev = self.device
        # launch fast path (the bench / training loop): same buffers, same modes, production draws -> only the seed changes in the cached struct
        no_draws = eps is None and model_idx is None and sel_noise is None and reset_idx is None and reset_model is None
        chunked = resume is not None or last_state is not None or stop is not None or t0 != 0 or stop_batch
        if no_draws and not chunked and out is not None and not force_generic and isinstance(pool, torch.Tensor):
            key = (id(out), pool.data_ptr(), B, T, H, sam_mode, bool(determ), bool(eval_all_heads), int(stream_offset))
            cached = getattr(self, '_ra_cache', None)
            if cached is not None and cached[0] == key:
                a = cached[1]
                a.seed = int(seed)
                self._chk(lib.metrpo_rollout(self._ctx, C.byref(a), self._stream()))
                return out
        pool = _f32(pool, dev); assert pool.dim() == 2 and pool.shape[1] == self.ns
        eps = _f32(eps, dev, (T, B, self.na)); sel_noise = _f32(sel_noise, dev, (T, B, self.ns))
        model_idx = _i32(model_idx, dev, (T, B))
        reset_idx = _i32(reset_idx, dev, (T + 1, B)); reset_model = _i32(reset_model, dev, (T + 1, B))
        if out is None:
            out = self.alloc_trajectory(B, T, H)
        a = _lib.RolloutArgs()
        a.B, a.T, a.H, a.sam_mode = B, T, H, _lib.SAM_MODES[sam_mode]
        a.determ, a.eval_all_heads = int(bool(determ)), int(bool(eval_all_heads))
        a.d_pool, a.n_pool, a.seed, a.stream_offset = pool.data_ptr(), pool.shape[0], int(seed), int(stream_offset)
        for name, t in (('d_eps', eps), ('d_model_idx', model_idx), ('d_sel_noise', sel_noise),
                        ('d_reset_idx', reset_idx), ('d_reset_model', reset_model)):
            setattr(a, name, t.data_ptr() if t is not None else None)
        a.d_obs, a.d_act, a.d_rew, a.d_mean = out.obs.data_ptr(), out.act.data_ptr(), out.rew.data_ptr(), out.mean.data_ptr()
        a.d_done, a.d_tpath, a.d_last_obs = out.done.data_ptr(), out.tpath.data_ptr(), out.last_obs.data_ptr()
        a.t0 = int(t0)
        if resume is not None:
            r_obs, r_ts, r_model = resume
            assert r_obs.dtype == torch.float32 and r_ts.dtype == torch.int32 and r_model.dtype == torch.int32
            assert tuple(r_obs.shape) == (B, self.ns) and r_ts.numel() == B and r_model.numel() == B
            a.d_init_obs, a.d_init_ts, a.d_init_model = r_obs.data_ptr(), r_ts.data_ptr(), r_model.data_ptr()
        if last_state is not None:
            assert all(t.dtype == torch.int32 and t.numel() == B for t in last_state)
            a.d_last_ts, a.d_last_model = last_state[0].data_ptr(), last_state[1].data_ptr()
        if stop is not None:
            assert stop.dtype == torch.int32
            a.d_stop = stop.data_ptr()
        if stop_batch:
            assert stop_cum is not None and stop_cum.dtype == torch.float64 and stop_cum.is_cuda
            a.stop_batch, a.d_stop_cum = int(stop_batch), stop_cum.data_ptr()
        fn = lib.metrpo_rollout_generic if force_generic else lib.metrpo_rollout
        self._chk(fn(self._ctx, C.byref(a), self._stream()))
        self._keep = (pool, eps, model_idx, sel_noise, reset_idx, reset_model, resume, last_state, stop, stop_cum)   # alive until the stream has consumed them
        if no_draws and not chunked and not force_generic:
            self._ra_cache = ((id(out), pool.data_ptr(), B, T, H, sam_mode, bool(determ), bool(eval_all_heads), int(stream_offset)), a, out, pool)
        return out

    def sampler_progress(self, done, tpath, t0, batch_size, counts, state, stop):
        """Loop condition of obtain_samples for the chunk (done, tpath [T,B]) that starts at global step t0; see metrpo.h."""
        T, B = done.shape
        assert counts.dtype == torch.float64 and counts.numel() >= T and state.dtype == torch.float64 and stop.dtype == torch.int32
        self._chk(lib.metrpo_sampler_progress(self._ctx, _ptr(done), _ptr(tpath), int(T), int(B), int(t0), int(batch_size),
                                              _ptr(counts), _ptr(state), _ptr(stop), self._stream()))

    def rollout_path(self):
        """3 step-wise GEMM (large nets), 2 cooperative-heads MFMA, 1 head-per-wave MFMA, 0 generic (current selection)."""
        return int(lib.metrpo_set_rollout_variant(self._ctx, int(getattr(self, '_variant', 0))))

    def last_rollout_kernel(self):
        """Kernel family the last rollout() of this engine ran on: 'generic', 'mfma-head-per-wave', 'mfma-cooperative', 'gemm-stepwise',
        'resident' (whole time loop in one launch, rollout_resident.hip), 'gemm-streamk' (step-wise, the whole ensemble of a step in one
        evenly split launch, mlp_streamk.h), 'streamk-persistent' (all steps of a chunk in one launch, rollout_persist.hip); None before the first rollout."""
        k = int(lib.metrpo_last_rollout_kernel(self._ctx))
        return {0: 'generic', 1: 'mfma-head-per-wave', 2: 'mfma-cooperative', 3: 'gemm-stepwise', 4: 'resident', 5: 'gemm-streamk', 6: 'streamk-persistent'}.get(k)

    def rollout_note(self):
        """Why the last rollout() ran outside the fast dispatch table (K != 5 at 2x64, hidden widths 65..127, ...); '' when it did not."""
        return lib.metrpo_rollout_note(self._ctx).decode()

    def alloc_trajectory(self, B, T, H):
        dev, f = self.device, torch.float32
        return Trajectory(torch.empty(T, B, self.ns, dtype=f, device=dev), torch.empty(T, B, self.na, dtype=f, device=dev),
                          torch.empty(T, B, dtype=f, device=dev), torch.empty(T, B, self.na, dtype=f, device=dev),
                          torch.empty(T, B, dtype=torch.uint8, device=dev), torch.empty(T, B, dtype=torch.int32, device=dev),
                          torch.empty(B, self.ns, dtype=f, device=dev), B, T, H)

    def validation_cost(self, s0, T, gamma):
        s0 = _f32(s0, self.device); assert s0.shape[1] == self.ns
        costs = torch.empty(self.K, dtype=torch.float64, device=self.device)
        self._chk(lib.metrpo_validation_cost(self._ctx, _ptr(s0), s0.shape[0], int(T), float(gamma), _ptr(costs), self._stream()))
        return costs

    # ------------------------------------------------------------------ process_samples pieces
    def gae(self, traj, coeffs, gamma, lam, stats=None):
        """-> adv (uncentred), ret, valid, stats[3] = (sum adv, sum adv^2, count) over valid samples."""
        dev = self.device
        T, B = traj.T, traj.B
        adv = torch.empty(T, B, dtype=torch.float32, device=dev); ret = torch.empty_like(adv)
        valid = torch.empty(T, B, dtype=torch.uint8, device=dev)
        if stats is None:
            stats = torch.zeros(3, dtype=torch.float64, device=dev)       # else: caller-provided, already zero
        if coeffs is not None:
            if isinstance(coeffs, torch.Tensor) and coeffs.is_cuda:
                coeffs = coeffs.to(torch.float64).contiguous()
            else:
                # host coefficients: pinned staging + async copy, so that this call does not block the host behind the rollout that is
                # still running on the stream (a pageable H2D copy would) and the GAE kernels are queued while it runs
                F = 2 * self.ns + 4
                if getattr(self, '_coef_pin', None) is None:
                    self._coef_pin = [torch.empty(F, dtype=torch.float64).pin_memory() for _ in range(2)]
                    self._coef_dev = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(2)]
                    self._coef_i = 0
                self._coef_i ^= 1
                pin, cd = self._coef_pin[self._coef_i], self._coef_dev[self._coef_i]
                pin.copy_(torch.as_tensor(np.asarray(coeffs, dtype=np.float64)))
                cd.copy_(pin, non_blocking=True)
                coeffs = cd
            assert coeffs.numel() == 2 * self.ns + 4
        self._chk(lib.metrpo_gae(self._ctx, _ptr(traj.obs), _ptr(traj.rew), _ptr(traj.done), _ptr(traj.tpath), T, B,
                                 _ptr(coeffs), float(gamma), float(lam), _ptr(adv), _ptr(ret), _ptr(valid), _ptr(stats),
                                 self._stream()))
        return adv, ret, valid, stats

    def process_begin(self, n_acc):
        """-> (old_log_std [na] float32, acc [n_acc] float64 zeros): metrpo_process_begin, one launch for what policy.log_std() + torch.zeros are as three.
        Closes an update that is only enqueued first (as get_policy does): the snapshot must see the policy the rollout ran with."""
        self._close_open_update()
        ls = torch.empty(self.na, dtype=torch.float32, device=self.device)
        acc = torch.empty(int(n_acc), dtype=torch.float64, device=self.device)
        self._chk(lib.metrpo_process_begin(self._ctx, _ptr(ls), _ptr(acc), int(n_acc), self._stream()))
        return ls, acc

    def center_advantages(self, adv, valid, stats):
        self._chk(lib.metrpo_center_advantages(self._ctx, _ptr(adv), _ptr(valid), adv.numel(), _ptr(stats), self._stream()))
        return adv

    def baseline_gram(self, obs, ret, tpath, valid, out=None):
        """out: optional zeroed float64 buffer of F*F + F elements (AtA row-major, then Aty) that receives the sums."""
        F = 2 * self.ns + 4
        if out is None:
            out = torch.zeros(F * F + F, dtype=torch.float64, device=self.device)
        AtA, Aty = out[:F * F].view(F, F), out[F * F:]
        self._chk(lib.metrpo_baseline_gram(self._ctx, _ptr(obs), _ptr(ret), _ptr(tpath), _ptr(valid), ret.numel(),
                                           _ptr(AtA), _ptr(Aty), self._stream()))
        return AtA, Aty

    # ------------------------------------------------------------------ TRPO update pieces
    def baseline_solve(self, gram, reg_coeff=1e-5, out=None):
        """LinearFeatureBaseline.fit's solve on the device: gram = [AtA (F*F, row-major) | Aty (F)] float64 device tensor (summed over the
        ranks) -> coefficients [F] float64 on the device, stream-ordered (no host round trip)."""
        F = 2 * self.ns + 4
        assert gram.is_cuda and gram.dtype == torch.float64 and gram.numel() == F * F + F and gram.is_contiguous()
        if out is None:
            out = torch.empty(F, dtype=torch.float64, device=self.device)
        self._chk(lib.metrpo_baseline_solve(self._ctx, _ptr(gram), C.c_void_p(gram.data_ptr() + 8 * F * F), float(reg_coeff), _ptr(out), self._stream()))
        return out

    def make_batch(self, obs, act, adv, old_mean, old_log_std, valid=None, n_global=None):
        dev = self.device
        obs = _f32(obs, dev).reshape(-1, self.ns); N = obs.shape[0]
        act = _f32(act, dev).reshape(N, self.na); adv = _f32(adv, dev).reshape(N)
        old_mean = _f32(old_mean, dev).reshape(N, self.na)
        old_log_std = _f32(old_log_std, dev)
        stride = 0 if old_log_std.numel() == self.na else self.na
        if stride:
            old_log_std = old_log_std.reshape(N, self.na)
        if valid is not None:
            valid = torch.as_tensor(valid, device=dev).to(torch.uint8).contiguous().reshape(N)
        if n_global is None:
            n_global = int(valid.sum().item()) if valid is not None else N
        b = _lib.Batch()
        b.d_obs, b.d_act, b.d_adv, b.d_old_mean = obs.data_ptr(), act.data_ptr(), adv.data_ptr(), old_mean.data_ptr()
        b.d_old_log_std, b.old_log_std_stride = old_log_std.data_ptr(), stride
        b.d_valid = valid.data_ptr() if valid is not None else None
        b.N, b.inv_n_global = N, 1.0 / float(n_global)
        b._keep = (obs, act, adv, old_mean, old_log_std, valid)
        return b

    def loss_grad(self, batch):
        out = torch.empty(self.P + 1, dtype=torch.float64, device=self.device)
        self._chk(lib.metrpo_loss_grad(self._ctx, C.byref(batch), _ptr(out), self._stream()))
        return out

    def fvp(self, batch, v):
        v = torch.as_tensor(v, device=self.device).to(torch.float64).contiguous()
        hv = torch.empty(self.P, dtype=torch.float64, device=self.device)
        self._chk(lib.metrpo_fvp(self._ctx, C.byref(batch), _ptr(v), _ptr(hv), self._stream()))
        return hv

    def loss_kl(self, batch, theta=None):
        theta = _f32(theta, self.device, (self.P,)) if theta is not None else None
        out = torch.empty(2, dtype=torch.float64, device=self.device)
        self._chk(lib.metrpo_loss_kl(self._ctx, C.byref(batch), _ptr(theta), _ptr(out), self._stream()))
        return out

    def trpo_update(self, batch, max_kl=0.01, cg_iters=10, reg_coeff=1e-5, backtrack_ratio=0.8, max_backtracks=15,
                    accept_violation=False, residual_tol=1e-10, allreduce=None, want_vectors=False, explicit_final_hvp=False, spec_trials=0):
        """One ConjugateGradientOptimizer.optimize; `allreduce(tensor_f64)` reduces in place across ranks.  spec_trials = S > 0: only
        enqueue it, with the first S line-search trials decided on the device (no synchronisation; see trpo_update_end)."""
        p = _lib.TrpoParams()
        p.max_kl, p.cg_iters, p.reg_coeff, p.backtrack_ratio = max_kl, cg_iters, reg_coeff, backtrack_ratio
        p.max_backtracks, p.accept_violation, p.residual_tol = max_backtracks, int(accept_violation), residual_tol
        p.explicit_final_hvp = int(bool(explicit_final_hvp))
        if allreduce is not None:
            dev = self.device

            def _cb(user, d_buf, count, stream):
                try:
                    view = torch.as_tensor(_DevView(d_buf, count), device=dev)
                    allreduce(view)
                    return 0
                except Exception:      # never let an exception cross the C ABI
                    import traceback
                    traceback.print_exc()
                    return -1
            cb = _lib.ALLREDUCE_FN(_cb)
            p.allreduce = cb
            self._cb_keepalive = cb
        diag = _lib.TrpoDiag()
        g = d = None
        if want_vectors:
            g = torch.empty(self.P, dtype=torch.float64, device=self.device); d = torch.empty_like(g)
        if spec_trials:
            # first half only (metrpo_trpo_update_begin): no synchronisation; trpo_update_end() returns the diagnostics
            self._chk(lib.metrpo_trpo_update_begin(self._ctx, C.byref(batch), C.byref(p), int(spec_trials), _ptr(g), _ptr(d), self._stream()))
            self._upd_open = (batch, p, g, d, int(spec_trials))              # keeps the batch struct and the output tensors alive until _end
            return None
        self._chk(lib.metrpo_trpo_update(self._ctx, C.byref(batch), C.byref(p), C.byref(diag), _ptr(g), _ptr(d), self._stream()))
        out = dict(loss_before=diag.loss_before, loss=diag.loss, kl=diag.kl, beta=diag.beta,
                   n_backtrack=diag.n_backtrack, accepted=bool(diag.accepted), cg_iters_run=diag.cg_iters_run)
        if want_vectors:
            out['g'], out['d'] = g, d
        return out

    def trpo_update_end(self):
        """Second half of trpo_update(..., spec_trials=S): waits for the update (not for launches enqueued after it) and returns its
        diagnostics; 'late' is True when the policy changed inside this call (accepted at a trial >= S): work enqueued since the first
        half saw the previous policy and must be redone."""
        if getattr(self, '_upd_open', None) is None:
            # already closed -- by get_policy() / set_policy(), which must not see or overwrite a policy whose search is still open
            closed = getattr(self, '_upd_closed', None)
            if closed is None:
                raise RuntimeError('trpo_update_end: no update is open')
            self._upd_closed = None
            return closed
        batch, p, g, d, spec = self._upd_open
        diag = _lib.TrpoDiag()
        late = C.c_int32(0)
        try:
            self._chk(lib.metrpo_trpo_update_end(self._ctx, C.byref(diag), C.byref(late), self._stream()))
        finally:
            self._upd_open = None
        out = dict(loss_before=diag.loss_before, loss=diag.loss, kl=diag.kl, beta=diag.beta,
                   n_backtrack=diag.n_backtrack, accepted=bool(diag.accepted), cg_iters_run=diag.cg_iters_run)
        out['late'] = bool(late.value)
        if g is not None:
            out['g'], out['d'] = g, d
        return out

    def _close_open_update(self):
        """Reading or replacing the policy while an update is only enqueued: close it first (its diagnostics wait for the optimizer's finish())."""
        if getattr(self, '_upd_open', None) is not None:
            self._upd_closed = self.trpo_update_end()


    # ------------------------------------------------------------------ ensemble dynamics training
    def get_dynamics(self):
        out = torch.empty(self.K, self.dyn_param_count, dtype=torch.float32, device=self.device)
        self._chk(lib.metrpo_get_dynamics(self._ctx, _ptr(out), self._stream()))
        return out

    def set_dynamics_model(self, k, params_k):
        p = _f32(params_k, self.device, (self.dyn_param_count,))
        self._chk(lib.metrpo_set_dynamics_model(self._ctx, int(k), _ptr(p), self._stream()))
        self._keep_model = p

    def set_normalizers(self, in_mean, in_std, diff_mean, diff_std):
        dev = self.device
        a = _f32(in_mean, dev, (self.ns + self.na,)); b = _f32(in_std, dev, (self.ns + self.na,))
        c = _f32(diff_mean, dev, (self.ns,)); e = _f32(diff_std, dev, (self.ns,))
        self._chk(lib.metrpo_set_normalizers(self._ctx, _ptr(a), _ptr(b), _ptr(c), _ptr(e), self._stream()))
        torch.cuda.current_stream(dev).synchronize()

    def train_reset(self):
        self._chk(lib.metrpo_dyn_train_reset(self._ctx, self._stream()))

    def train_step(self, x, y, batch_size, lr, reg_constant=0.0, beta1=0.9, beta2=0.999, eps=1e-8, want_loss=True):
        """x [batch_size*K, ns+na], y [batch_size*K, ns] device tensors; returns per-model losses (before the update)."""
        dev = self.device
        x = _f32(x, dev, (batch_size * self.K, self.ns + self.na)); y = _f32(y, dev, (batch_size * self.K, self.ns))
        tp = _lib.TrainParams(lr, beta1, beta2, eps, reg_constant, batch_size)
        loss = torch.empty(self.K, dtype=torch.float64, device=dev) if want_loss else None
        self._chk(lib.metrpo_dyn_train_step(self._ctx, _ptr(x), _ptr(y), C.byref(tp), _ptr(loss), self._stream()))
        self._keep_train = (x, y)
        return loss

    def eval_losses(self, x, y, reg_constant=0.0):
        dev = self.device
        x = _f32(x, dev); y = _f32(y, dev, (x.shape[0], self.ns))
        out = torch.empty(self.K, dtype=torch.float64, device=dev)
        self._chk(lib.metrpo_dyn_eval_losses(self._ctx, _ptr(x), _ptr(y), x.shape[0], float(reg_constant), _ptr(out), self._stream()))
        self._keep_eval = (x, y)
        return out

    def rms_accumulate(self, x, rsum, rsumsq):
        x = _f32(x, self.device)
        self._chk(lib.metrpo_rms_accumulate(self._ctx, _ptr(x), x.shape[0], x.shape[1], _ptr(rsum), _ptr(rsumsq), self._stream()))
        self._keep_rms = x

    def set_det_path(self, use_mfma):
        """test hook: per-model deterministic rollouts (validation cost, BPTT sweeps) on the MFMA kernels (True) or the generic ones."""
        return int(lib.metrpo_set_det_path(self._ctx, 1 if use_mfma else 0))

    # ---- BPTT policy update ('bptt' branch of optimize_policy, model_based_rl.py:1181-1187) ----
    def bptt_grad(self, init_states, T, gamma=1.0):
        """-> (costs [K] float64 device tensor = policy_costs per model, grad [P] float64 device tensor of mean_k cost)."""
        x0 = _f32(init_states, self.device)
        costs = torch.empty(self.K, dtype=torch.float64, device=self.device)
        grad = torch.empty(self.P, dtype=torch.float64, device=self.device)
        self._chk(lib.metrpo_bptt_grad(self._ctx, _ptr(x0), x0.shape[0], int(T), float(gamma), _ptr(costs), _ptr(grad), self._stream()))
        self._keep_bptt = x0
        return costs, grad

    def policy_adam_reset(self):
        self._chk(lib.metrpo_policy_adam_reset(self._ctx, self._stream()))

    def policy_adam_step(self, grad, lr, clip_val=None, beta1=0.9, beta2=0.999, eps=1e-8):
        g = grad if isinstance(grad, torch.Tensor) else torch.as_tensor(grad)
        g = g.to(self.device, torch.float64).contiguous()
        self._chk(lib.metrpo_policy_adam_step(self._ctx, _ptr(g), float(lr), float(beta1), float(beta2), float(eps),
                                              float(clip_val) if clip_val else 0.0, self._stream()))
        self._keep_adam = g


class _DevView(object):
    """Zero-copy float64 view of library-owned device memory for torch (CUDA array interface)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': '<f8', 'data': (int(ptr), False), 'version': 2}


def xavier_policy_theta(ns, hidden, na, init_std=1.0, seed=0):
    """[rllab] GaussianMLPPolicy initial parameters: Xavier-uniform W, zero b, log_std = log(init_std)."""
    rng = np.random.RandomState(seed)
    dims = [ns] + list(hidden) + [na]
    parts = []
    for i in range(len(dims) - 1):
        lim = np.sqrt(6.0 / (dims[i] + dims[i + 1]))
        parts += [rng.uniform(-lim, lim, size=dims[i] * dims[i + 1]), np.zeros(dims[i + 1])]
    parts.append(np.full(na, np.log(init_std)))
    return np.concatenate(parts).astype(np.float32)
