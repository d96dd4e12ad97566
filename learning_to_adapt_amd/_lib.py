"""ctypes binding of ``libl2a_hip.so`` (C ABI: ``include/l2a.h``).

There is deliberately no fallback: if the shared library is missing or a call fails, an
exception is raised.  The CPU oracle under ``oracle/`` is test infrastructure and is never
imported from here.
"""

import ctypes
import json
import os

from .envs.reward_spec import RewardSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
# L2A_LIB_PATH: developer override for kernel A/B runs (tools/build_variant.py); the product loads the in-tree build
LIB_PATH = os.environ.get("L2A_LIB_PATH") or os.path.join(_HERE, "libl2a_hip.so")

# include/l2a.h
L2A_OK = 0
L2A_ESPLIT = -5
L2A_STEP_MISS = 1
L2A_STEP_UNSPLIT = 2
L2A_STEP_DREW = 3
ACT_CODES = {None: 0, "identity": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "swish": 4}
MODE_CODES = {"single": 0, "per_block": 1, "mean": 2}
KERNEL_CODES = {"auto": 0, "mfma": 1, "valu": 2}
CELL_CODES = {"lstm": 0, "gru": 1, "rnn": 2}

EXPORTED_SYMBOLS = (
    "l2a_init", "l2a_destroy", "l2a_last_error", "l2a_device_info", "l2a_set_kernel", "l2a_set_split", "l2a_set_batch", "l2a_set_xcd_align", "l2a_set_fan", "l2a_set_double_rounds", "l2a_set_micro",
    "l2a_launch_status", "l2a_set_debug_buffer", "l2a_set_spin_limit", "l2a_inject_status",
    "l2a_model_create", "l2a_model_destroy", "l2a_model_set_weights", "l2a_model_set_weights_strided",
    "l2a_model_set_norm", "l2a_model_adapt_sgd", "l2a_model_adapt_sgd_host", "l2a_model_adapt_sgd_raw", "l2a_model_get_weights",
    "l2a_plan_rs", "l2a_plan_rs_sync", "l2a_plan_rs_chunk", "l2a_predict", "l2a_key_encode", "l2a_key_decode", "l2a_mfma_eligible", "l2a_plan_geometry",
    "l2a_packed_layer_floats", "l2a_pack_layer_host", "l2a_micro_layout_floats", "l2a_micro_pack_layer_host",
    "l2a_comm_unique_id", "l2a_comm_init", "l2a_comm_destroy", "l2a_allreduce_best", "l2a_plan_payload",
    "l2a_cem_sample", "l2a_cem_refit", "l2a_cem_pick",
    "l2a_lstm_create", "l2a_rnn_create", "l2a_lstm_destroy", "l2a_lstm_set_weights", "l2a_lstm_set_norm", "l2a_lstm_plan_rs", "l2a_lstm_plan_rs_sync", "l2a_lstm_plan_rs_chunk",
    "l2a_lstm_predict", "l2a_lstm_advance", "l2a_lstm_mfma_eligible",
    "l2a_controller_create", "l2a_controller_create_sharded", "l2a_controller_create_sharded_device", "l2a_lstm_controller_create", "l2a_controller_create_device", "l2a_lstm_controller_create_device",
    "l2a_controller_destroy", "l2a_controller_step", "l2a_controller_begin", "l2a_lstm_controller_begin", "l2a_controller_finish",
    "l2a_lstm_controller_step", "l2a_controller_rearm", "l2a_controller_actions", "l2a_controller_stats",
)


class L2AError(RuntimeError):
    pass


def plan_geometry(obs_dim, act_dim, hidden, n_sets, mode, m, n, h, split=-1, fan=-1, micro=-1, cus=0, double=-1):
    """The launch geometry the library would pick for this plan (``l2a_plan_geometry``: the launcher's own decision code, no
    GPU needed).  Returns a dict: kernel ('valu' | 'mfma16' | 'micro'), nt, split (0 none, 1 whole sets, 2 shared half member,
    3 member fan), split_from, fan, workgroups, lds_bytes, sets_per_batch, micro_tiles, placement_units, front_workgroups
    (double-tile workgroups of a launch IN FRONT of the described one: ``l2a_set_double_rounds``), whole_instance (the
    described launch runs on a whole-tiles-only kernel instance)."""
    lib = load()
    hid = (ctypes.c_int * len(hidden))(*[int(x) for x in hidden])
    pol = (ctypes.c_int * 5)(int(split), int(fan), int(micro), int(cus), int(double))
    out = (ctypes.c_int * 12)()
    rc = lib.l2a_plan_geometry(int(obs_dim), int(act_dim), len(hidden), hid, int(n_sets), MODE_CODES[mode], int(m), int(n), int(h), pol, out)
    if rc != L2A_OK:
        raise L2AError("l2a_plan_geometry failed (%d)" % rc)
    v = list(out)
    return dict(kernel=("valu", "mfma16", "micro")[v[0]], nt=v[1], split=v[2], split_from=v[3], fan=bool(v[4]), workgroups=v[5],
                lds_bytes=v[6], sets_per_batch=v[7], micro_tiles=v[8], placement_units=v[9], front_workgroups=v[10],
                whole_instance=bool(v[11]))


_lib = None


def load():
    """Load the shared library (once) and declare every prototype of include/l2a.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise L2AError(
            "%s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, i32, f32 = c.c_void_p, c.c_int, c.c_float
    lib.l2a_init.argtypes = [i32, c.POINTER(vp)]
    lib.l2a_init.restype = i32
    lib.l2a_destroy.argtypes = [vp]
    lib.l2a_destroy.restype = None
    lib.l2a_last_error.argtypes = [vp]
    lib.l2a_last_error.restype = c.c_char_p
    lib.l2a_device_info.argtypes = [vp, c.c_char_p, i32]
    lib.l2a_device_info.restype = i32
    lib.l2a_set_kernel.argtypes = [vp, i32]
    lib.l2a_set_kernel.restype = i32
    lib.l2a_set_split.argtypes = [vp, i32]
    lib.l2a_set_split.restype = i32
    if os.environ.get("L2A_LIB_PATH") and not hasattr(lib, "l2a_set_batch"):
        lib.l2a_set_batch = lambda handle, sets: 0         # developer A/B against a library of an earlier round
    else:
        lib.l2a_set_batch.argtypes = [vp, i32]
        lib.l2a_set_batch.restype = i32
    if os.environ.get("L2A_LIB_PATH") and not hasattr(lib, "l2a_set_micro"):
        lib.l2a_set_micro = lambda handle, policy: 0
    else:
        lib.l2a_set_micro.argtypes = [vp, i32]
        lib.l2a_set_micro.restype = i32
    if os.environ.get("L2A_LIB_PATH") and not hasattr(lib, "l2a_set_fan"):
        lib.l2a_set_fan = lambda handle, on: 0
    else:
        lib.l2a_set_fan.argtypes = [vp, i32]
        lib.l2a_set_fan.restype = i32
    if os.environ.get("L2A_LIB_PATH") and not hasattr(lib, "l2a_set_double_rounds"):
        lib.l2a_set_double_rounds = lambda handle, on: 0
    else:
        lib.l2a_set_double_rounds.argtypes = [vp, i32]
        lib.l2a_set_double_rounds.restype = i32
    if os.environ.get("L2A_LIB_PATH") and not hasattr(lib, "l2a_set_xcd_align"):
        lib.l2a_set_xcd_align = lambda handle, on: 0
    else:
        lib.l2a_set_xcd_align.argtypes = [vp, i32]
        lib.l2a_set_xcd_align.restype = i32
    lib.l2a_launch_status.argtypes = [vp, c.POINTER(i32)]
    lib.l2a_launch_status.restype = i32
    lib.l2a_set_debug_buffer.argtypes = [vp, vp]
    lib.l2a_set_debug_buffer.restype = i32
    lib.l2a_set_spin_limit.argtypes = [vp, c.c_uint]
    lib.l2a_set_spin_limit.restype = i32
    lib.l2a_inject_status.argtypes = [vp, i32]
    lib.l2a_inject_status.restype = i32
    lib.l2a_model_create.argtypes = [vp, i32, i32, i32, c.POINTER(i32), i32, i32, i32, i32, c.POINTER(vp)]
    lib.l2a_model_create.restype = i32
    lib.l2a_model_destroy.argtypes = [vp]
    lib.l2a_model_destroy.restype = None
    lib.l2a_model_set_weights.argtypes = [vp, i32, c.POINTER(vp), vp]
    lib.l2a_model_set_weights.restype = i32
    lib.l2a_model_set_weights_strided.argtypes = [vp, i32, i32, c.POINTER(vp), c.POINTER(c.c_longlong), vp]
    lib.l2a_model_set_weights_strided.restype = i32
    lib.l2a_model_adapt_sgd.argtypes = [vp, c.POINTER(vp), vp, vp, i32, i32, f32, vp]
    lib.l2a_model_adapt_sgd.restype = i32
    lib.l2a_model_adapt_sgd_host.argtypes = [vp, c.POINTER(vp), vp, vp, i32, i32, f32, vp]
    lib.l2a_model_adapt_sgd_host.restype = i32
    lib.l2a_model_adapt_sgd_raw.argtypes = [vp, c.POINTER(vp), vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]
    lib.l2a_model_adapt_sgd_raw.restype = i32
    lib.l2a_model_get_weights.argtypes = [vp, i32, c.POINTER(vp), vp]
    lib.l2a_model_get_weights.restype = i32
    dp = c.POINTER(c.c_double)
    lib.l2a_model_set_norm.argtypes = [vp, i32, dp, dp, dp, dp, dp, dp, vp]
    lib.l2a_model_set_norm.restype = i32
    lib.l2a_plan_rs.argtypes = [vp, vp, vp, i32, i32, i32, c.c_double, c.POINTER(RewardSpec), i32, vp, vp, vp]
    lib.l2a_plan_rs.restype = i32
    lib.l2a_plan_rs_sync.argtypes = [vp, vp, vp, i32, i32, i32, c.c_double, c.POINTER(RewardSpec), i32, vp, vp, vp]
    lib.l2a_plan_rs_sync.restype = i32
    lib.l2a_plan_rs_chunk.argtypes = [vp, vp, i32, vp, i32, i32, i32, i32, c.c_double, c.POINTER(RewardSpec), i32, vp, vp, vp, vp, vp]
    lib.l2a_plan_rs_chunk.restype = i32
    lib.l2a_predict.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.l2a_predict.restype = i32
    lib.l2a_key_encode.argtypes = [f32, i32]
    lib.l2a_key_encode.restype = c.c_ulonglong
    lib.l2a_key_decode.argtypes = [c.c_ulonglong, c.POINTER(f32), c.POINTER(i32)]
    lib.l2a_key_decode.restype = None
    lib.l2a_mfma_eligible.argtypes = [i32, i32, i32, c.POINTER(i32)]
    lib.l2a_mfma_eligible.restype = i32
    lib.l2a_plan_geometry.argtypes = [i32, i32, i32, c.POINTER(i32), i32, i32, i32, i32, i32, c.POINTER(i32), c.POINTER(i32)]
    lib.l2a_plan_geometry.restype = i32
    lib.l2a_packed_layer_floats.argtypes = [i32, i32]
    lib.l2a_packed_layer_floats.restype = c.c_longlong
    lib.l2a_pack_layer_host.argtypes = [c.POINTER(f32), i32, i32, c.POINTER(f32)]
    lib.l2a_pack_layer_host.restype = i32
    if hasattr(lib, "l2a_micro_layout_floats"):         # (absent from older variant libraries selected with L2A_LIB_PATH)
        lib.l2a_micro_layout_floats.argtypes = [i32, i32, i32, i32]
        lib.l2a_micro_layout_floats.restype = c.c_longlong
        lib.l2a_micro_pack_layer_host.argtypes = [c.POINTER(f32), i32, i32, i32, i32, i32, c.POINTER(f32)]
        lib.l2a_micro_pack_layer_host.restype = i32
    lib.l2a_comm_unique_id.argtypes = [c.c_char_p]
    lib.l2a_comm_unique_id.restype = i32
    lib.l2a_comm_init.argtypes = [vp, i32, i32, c.c_char_p]
    lib.l2a_comm_init.restype = i32
    lib.l2a_comm_destroy.argtypes = [vp]
    lib.l2a_comm_destroy.restype = i32
    lib.l2a_allreduce_best.argtypes = [vp, vp, i32, vp]
    lib.l2a_allreduce_best.restype = i32
    lib.l2a_lstm_create.argtypes = [vp, i32, i32, i32, i32, i32, c.POINTER(vp)]
    lib.l2a_lstm_create.restype = i32
    lib.l2a_rnn_create.argtypes = [vp, i32, i32, i32, c.POINTER(i32), i32, i32, i32, c.POINTER(vp)]
    lib.l2a_rnn_create.restype = i32
    lib.l2a_lstm_destroy.argtypes = [vp]
    lib.l2a_lstm_destroy.restype = None
    lib.l2a_lstm_set_weights.argtypes = [vp, c.POINTER(vp), vp]
    lib.l2a_lstm_set_weights.restype = i32
    lib.l2a_lstm_set_norm.argtypes = [vp, dp, dp, dp, dp, dp, dp, vp]
    lib.l2a_lstm_set_norm.restype = i32
    lib.l2a_lstm_plan_rs.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, c.c_double, c.POINTER(RewardSpec), i32, vp, vp, vp]
    lib.l2a_lstm_plan_rs.restype = i32
    lib.l2a_cem_sample.argtypes = [vp, vp, c.c_ulonglong, c.c_ulonglong, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32,
                                   vp, vp, vp, vp]
    lib.l2a_cem_sample.restype = i32
    lib.l2a_cem_refit.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp]
    lib.l2a_cem_refit.restype = i32
    lib.l2a_cem_pick.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.l2a_cem_pick.restype = i32
    lib.l2a_lstm_plan_rs_sync.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, c.c_double, c.POINTER(RewardSpec), i32, vp, vp, vp, vp]
    lib.l2a_lstm_plan_rs_sync.restype = i32
    lib.l2a_lstm_plan_rs_chunk.argtypes = [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, c.c_double, c.POINTER(RewardSpec), i32,
                                           vp, vp, vp, vp, vp, vp, vp]
    lib.l2a_lstm_plan_rs_chunk.restype = i32
    lib.l2a_lstm_predict.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    lib.l2a_lstm_predict.restype = i32
    if hasattr(lib, "l2a_lstm_advance"):
        lib.l2a_lstm_advance.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp]
        lib.l2a_lstm_advance.restype = i32
    if os.environ.get("L2A_LIB_PATH") and not hasattr(lib, "l2a_plan_payload"):
        pass                                               # developer A/B against a library of an earlier round
    else:
        lib.l2a_plan_payload.argtypes = [vp, vp, i32, c.c_ulonglong, vp, vp]
        lib.l2a_plan_payload.restype = i32
    lib.l2a_lstm_mfma_eligible.argtypes = [i32, i32, i32]
    lib.l2a_lstm_mfma_eligible.restype = i32
    if os.environ.get("L2A_LIB_PATH") and not hasattr(lib, "l2a_controller_step"):
        pass                                               # developer A/B against a library of an earlier round
    else:
        lib.l2a_controller_create.argtypes = [vp, i32, i32, i32, vp, vp, c.c_double, c.POINTER(RewardSpec), vp, i32, c.POINTER(vp)]
        # the caller's collective of a sharded step: int fn(void* arg, uint64* payload_dev, int words, void* stream)
        lib.REDUCE_FN = c.CFUNCTYPE(i32, vp, vp, i32, vp)
        lib.l2a_controller_create.restype = i32
        lib.l2a_lstm_controller_create.argtypes = lib.l2a_controller_create.argtypes
        lib.l2a_lstm_controller_create.restype = i32
        lib.l2a_controller_create_device.argtypes = [vp, i32, i32, i32, vp, vp, c.c_double, c.POINTER(RewardSpec), c.c_ulonglong, c.POINTER(vp)]
        lib.l2a_controller_create_device.restype = i32
        lib.l2a_lstm_controller_create_device.argtypes = lib.l2a_controller_create_device.argtypes
        lib.l2a_lstm_controller_create_device.restype = i32
        lib.l2a_controller_destroy.argtypes = [vp]
        lib.l2a_controller_destroy.restype = None
        lib.l2a_controller_step.argtypes = [vp, vp, vp, vp, vp, vp]
        lib.l2a_controller_step.restype = i32
        lib.l2a_lstm_controller_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.l2a_lstm_controller_step.restype = i32
        lib.l2a_controller_rearm.argtypes = [vp]
        lib.l2a_controller_rearm.restype = i32
        lib.l2a_controller_create_sharded.argtypes = [vp, i32, i32, i32, vp, vp, c.c_double, c.POINTER(RewardSpec), vp, i32, i32, i32,
                                                      vp, vp, c.POINTER(vp)]
        lib.l2a_controller_create_sharded.restype = i32
        lib.l2a_controller_create_sharded_device.argtypes = [vp, i32, i32, i32, vp, vp, c.c_double, c.POINTER(RewardSpec), c.c_ulonglong, i32, i32,
                                                             vp, vp, c.POINTER(vp)]
        lib.l2a_controller_create_sharded_device.restype = i32
        lib.l2a_controller_begin.argtypes = [vp, vp, vp]
        lib.l2a_controller_begin.restype = i32
        lib.l2a_lstm_controller_begin.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        lib.l2a_lstm_controller_begin.restype = i32
        lib.l2a_controller_finish.argtypes = [vp, vp, vp, vp]
        lib.l2a_controller_finish.restype = i32
        lib.l2a_controller_actions.argtypes = [vp]
        lib.l2a_controller_actions.restype = vp
        lib.l2a_controller_stats.argtypes = [vp, dp, i32]
        lib.l2a_controller_stats.restype = i32
    _lib = lib
    return lib


def key_decode(key):
    """``best_key`` -> (return as float, global candidate index)."""
    lib = load()
    ret, idx = ctypes.c_float(), ctypes.c_int()
    lib.l2a_key_decode(ctypes.c_ulonglong(int(key) & 0xFFFFFFFFFFFFFFFF), ctypes.byref(ret), ctypes.byref(idx))
    return ret.value, idx.value


class Context(object):
    """One ``l2a_ctx`` per (process, device).  Created lazily by the models (never at import
    or construction time: the reference forks env workers before touching the accelerator,
    ``samplers/sampler.py:37`` vs ``trainers/mb_trainer.py:46-48``)."""

    _by_device = {}

    def __init__(self, device=0):
        self.lib = load()
        handle = ctypes.c_void_p()
        rc = self.lib.l2a_init(int(device), ctypes.byref(handle))
        if rc != L2A_OK:
            raise L2AError("l2a_init(device=%d) failed (%d): %s" % (
                device, rc, self.lib.l2a_last_error(None).decode()))
        self.handle = handle
        self.device = int(device)
        self.pid = os.getpid()

    @classmethod
    def get(cls, device=0):
        ctx = cls._by_device.get(device)
        if ctx is None or ctx.pid != os.getpid():
            ctx = cls(device)
            cls._by_device[device] = ctx
            kind = os.environ.get("L2A_KERNEL")
            if kind:
                ctx.set_kernel(kind)
        return ctx

    def check(self, rc, what):
        if rc != L2A_OK:
            raise L2AError("%s failed (%d): %s" % (what, rc, self.lib.l2a_last_error(self.handle).decode()))

    def info(self):
        buf = ctypes.create_string_buffer(512)
        self.check(self.lib.l2a_device_info(self.handle, buf, 512), "l2a_device_info")
        return json.loads(buf.value.decode())

    def set_kernel(self, kind):
        self.check(self.lib.l2a_set_kernel(self.handle, KERNEL_CODES[kind]), "l2a_set_kernel")

    def set_split(self, policy):
        """0 = never, 1 = auto (default; shares the middle set of odd ensembles), 2 = whole sets only."""
        self.check(self.lib.l2a_set_split(self.handle, int(policy)), "l2a_set_split")

    def set_batch(self, sets):
        """Sets per batch of the MFMA rollout: 0 = as many as fit the LDS (default), 1 = one at a time (bit-identical)."""
        self.check(self.lib.l2a_set_batch(self.handle, int(sets)), "l2a_set_batch")

    def set_micro(self, policy):
        """Micro-tile kernels: 0 = never, 1 = where they fill the chip better (default), 2 = whenever eligible.  Bit-identical to
        the 16-candidate matrix-core kernels; a generic recurrent stack too large for those (policy 0: VALU kernel) agrees to the
        fp32 tolerance only (include/l2a.h)."""
        self.check(self.lib.l2a_set_micro(self.handle, int(policy)), "l2a_set_micro")

    def set_xcd_align(self, on):
        """Split launches: ensemble group A on XCDs 0-3, B on 4-7 exactly (padded grid; default on; bit-identical)."""
        self.check(self.lib.l2a_set_xcd_align(self.handle, int(bool(on))), "l2a_set_xcd_align")

    def set_fan(self, on):
        """Member fan: small mean-ensemble plans on one workgroup per (candidate tile, member) - default on; bit-identical."""
        self.check(self.lib.l2a_set_fan(self.handle, int(bool(on))), "l2a_set_fan")

    def set_double_rounds(self, on):
        """Double rounds: multi-round plans at width 512 run their first rounds on two-tile workgroups - default on; bit-identical."""
        self.check(self.lib.l2a_set_double_rounds(self.handle, int(bool(on))), "l2a_set_double_rounds")

    def launch_status_value(self):
        """Status word of the launches since the last call (stream must be synchronised); reading clears it."""
        st = ctypes.c_int()
        self.check(self.lib.l2a_launch_status(self.handle, ctypes.byref(st)), "l2a_launch_status")
        return st.value

    def launch_status(self):
        """Status of the launches since the last call (stream must be synchronised).  Raises when a
        member-split exchange timed out - results of those launches are invalid."""
        st = self.launch_status_value()
        if st != 0:
            raise L2AError("rollout launch reported status 0x%x (member-split exchange timed out; "
                           "set L2A_SPLIT=0 to disable the split)" % st)
        return st

    def check_or_degrade(self):
        """After a stream sync: True when every launch since the last check was fine.  A tile-split exchange
        that timed out (the two workgroups of a tile were not co-resident, e.g. another process shares the GPU)
        invalidates those launches: the context is switched to the unsplit launch geometry for good - same
        bits, fewer busy CUs - and False is returned so that the caller relaunches.  Raises only when launches
        fail although the split is already off."""
        st = self.launch_status_value()
        if st == 0:
            return True
        if getattr(self, "split_degraded", False):
            raise L2AError("rollout launch reported status 0x%x with the tile split disabled" % st)
        self.set_split(0)
        self.split_degraded = True
        return False

    def force_unsplit(self):
        """A rank of a sharded plan was told by the collective that SOME rank's launch lost its tile-split partner: all
        ranks clear their status word, switch to the unsplit geometry (bit-identical results) and repeat the launch
        together.  Never raises: a rank that had already degraded on its own (a single-GPU call between two sharded
        plans) must stay in step with the others - a launch that fails again is caught by the caller's second look at the
        REDUCED flag, which raises on every rank alike."""
        self.launch_status_value()
        if not getattr(self, "split_degraded", False):
            self.set_split(0)
            self.split_degraded = True

    def plan_payload(self, best_key, m, digest, payload, stream_ptr):
        """Pack what a sharded plan all-reduces (``l2a_plan_payload``): keys, this rank's launch flag, the digest pair -
        on the device, in stream order behind the launch, no host synchronisation.  ``best_key`` / ``payload``: int64 CUDA
        tensors of ``m`` / ``m + 3`` elements.  Shared by every planner model (MLP and recurrent)."""
        assert best_key.is_cuda and payload.is_cuda and str(best_key.dtype) == "torch.int64" and str(payload.dtype) == "torch.int64"
        assert best_key.numel() == m and payload.numel() == m + 3
        rc = self.lib.l2a_plan_payload(self.handle, ctypes.c_void_p(best_key.data_ptr()), int(m), ctypes.c_ulonglong(int(digest)),
                                       ctypes.c_void_p(payload.data_ptr()), stream_ptr)
        self.check(rc, "l2a_plan_payload")

    def set_spin_limit(self, polls):
        """Developer / test knob: polls a split workgroup waits for its partner per launch (0 = default)."""
        self.check(self.lib.l2a_set_spin_limit(self.handle, int(polls)), "l2a_set_spin_limit")
