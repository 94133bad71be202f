"""ctypes binding of libmsl3d_hip.so (C ABI declared in include/mslesions3d_hip.h).

There is NO fallback: if the library is missing or a kernel launch fails, the caller gets an exception.
torch is only used by callers for device memory and streams; no torch type crosses this boundary.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsl3d_hip.so")

_P, _I, _F, _D, _Z, _Q = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_size_t, ctypes.c_longlong

# name -> (restype, argtypes); mirrors include/mslesions3d_hip.h one to one
_SIGNATURES = {
    "msl_abi_version": (_I, []),
    "msl_bn_finalize": (_I, [_P, _I, _D, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _P]),
    "msl_bn_finalize_entry_bytes": (_Z, []),
    "msl_bn_finalize_table_set": (_I, [_P, _I, _I, _P, _I, _D, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I]),
    "msl_bn_finalize_batch": (_I, [_P, _I, _I, _P]),
    "msl_bn_eval_affine_batch": (_I, [_P, _I, _I, _P]),
    "msl_bn_eval_affine": (_I, [_P, _P, _P, _P, _F, _P, _P, _I, _P]),
    "msl_bn_relu_materialize": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "msl_bn_relu_materialize_fold": (_I, [_P, _P, _I, _D, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _P]),
    "msl_bn_relu_bwd_num_partials": (_I, [_I, _I]),
    "msl_bn_relu_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "msl_bn_bwd_finalize": (_I, [_P, _I, _D, _P, _P, _P, _P, _I, _P]),
    "msl_bn_bwd_finalize_coef": (_I, [_P, _I, _D, _P, _P, _P, _I, _P]),
    "msl_bn_relu_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "msl_bn_relu_bwd_finalize_apply": (_I, [_P, _I, _D, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "msl_bn_relu_bwd_fused": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "msl_block_bwd_channel_link_supported": (_I, [_I, _I, _I, _I, _I]),
    "msl_pwconv_bwd_fused_num_partials": (_I, [_I, _I, _I, _I]),
    "msl_pwconv_bwd_fused": (_I, [_P, _P, _P, _P, _I, _D, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_block_bwd_channel_link": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_stem_conv_fwd_num_partials": (_I, [_I, _I, _I, _I]),
    "msl_stem_conv_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_stem_conv_bwd_weight_workspace_bytes": (_Z, [_I]),
    "msl_stem_conv_bwd_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_stem_conv_bwd_weight_bnapply": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_stem_conv_bwd_weight_fused": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_s2_bwd_bnreduce_bww_num_partials": (_I, [_I, _I, _I, _I, _I]),
    "msl_dwconv_s2_bwd_bnreduce_bww": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_s2_bwd_data_bnreduce_bww": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_bwd_weight_finalize": (_I, [_P, _I, _P, _I, _P]),
    "msl_dwconv_fwd_num_partials": (_I, [_I, _I, _I, _I, _I, _I]),
    "msl_dwconv_fwd_variant": (_I, [_I, _I, _I, _I, _I, _I]),
    "msl_dwconv_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_s1_bwd_data_resident": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_fwd_fold": (_I, [_P, _P, _I, _D, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_bwd_data_bnreduce_num_partials": (_I, [_I, _I, _I, _I, _I]),
    "msl_dwconv_bwd_data_bnreduce": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_bwd_weight_tiled": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_bwd_weight_tiled_num_partials": (_I, [_I, _I, _I, _I, _I, _I]),
    "msl_dwconv_bwd_weight_num_partials": (_I, [_I, _I, _I, _I, _I, _I]),
    "msl_dwconv_bwd_weight": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_pwconv_fwd_num_partials": (_I, [_I, _I, _I, _I]),
    "msl_pwconv_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_pwconv_fwd_fold": (_I, [_P, _P, _I, _D, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_pwconv_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_pwconv_bwd_weight_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "msl_pwconv_bwd_weight": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_pwconv_bwd_weight_nslabs": (_I, [_I, _I, _I, _I]),
    "msl_pwconv_bwd_weight_slabs": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_pwconv_bwd_weight_batchable": (_I, [_I, _I, _I, _I]),
    "msl_pwconv_bwd_weight_slabs_batch": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "msl_stem_conv_bwd_weight_nslabs": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "msl_head_conv_bwd_weight_nslabs": (_I, [_I, _I, _I, _I, _I, _I]),
    "msl_grad_reduce_entry_bytes": (_Z, []),
    "msl_grad_reduce_table_set": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _Q, _I, _I, _I]),
    "msl_grad_reduce_batch": (_I, [_P, _I, _I, _P]),
    "msl_grad_reduce_batch_indexed": (_I, [_P, _I, _P, _I, _P]),
    "msl_head_packed_weight_elems": (_Z, [_I, _I]),
    "msl_head_pack_weights": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "msl_head_pack_weights_batch": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "msl_head_fwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "msl_head_conv_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_head_grad_pack": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_head_grad_pack_batch": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_head_conv_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_head_bwd_weight_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "msl_head_conv_bwd_weight": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_stem_conv_fwd_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_fwd_bf16_num_partials": (_I, [_I, _I, _I, _I, _I, _I]),
    "msl_dwconv_wave_num_partials": (_I, [_I] * 6),
    "msl_dwconv_fwd_eval_rows_ok": (_I, [_I, _I, _I, _I, _I, _I]),
    "msl_dwconv_fwd_wave_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_bwd_data_s2_patch_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_fwd_wave_bf16_fold": (_I, [_P, _P, _I, _D, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_pwconv_fwd_bf16_fold": (_I, [_P, _P, _I, _D, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_dwconv_s2_bwd_bnreduce_bww_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "msl_stem_conv_bwd_weight_fused_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_bwd_weight_wave_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_fwd_small_eval_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_fwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_pwconv_fwd_bf16_num_partials": (_I, [_I, _I]),
    "msl_pwconv_fwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_bn_relu_materialize_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "msl_bn_relu_materialize_bf16_pad32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "msl_bn_relu_materialize_bf16_pad32_fold": (_I, [_P, _P, _I, _D, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P]),
    "msl_head_packed_weight_bf16_elems": (_Z, [_I]),
    "msl_head_pack_weights_bf16": (_I, [_P, _P, _P, _I, _I, _P]),
    "msl_head_conv_fwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_pwconv_bwd_data_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_pwconv_bwd_weight_bf16_nslabs": (_I, [_I, _I, _I, _I]),
    "msl_pwconv_bwd_weight_slabs_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_dwconv_bwd_data_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_dwconv_bwd_weight_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_bn_relu_bwd_bf16_num_partials": (_I, [_I, _I]),
    "msl_bn_relu_bwd_reduce_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "msl_bn_relu_bwd_apply_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "msl_bn_relu_bwd_fused_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "msl_block_bwd_channel_link_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_head_conv_bwd_data_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_head_conv_bwd_weight_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "msl_stem_conv_bwd_weight_bnapply_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "msl_make_priors": (_I, [_P, _I, _I, _I, _I, _D, _I, _P]),
    "msl_box_transform": (_I, [_P, _P, _P, _I, _I, _P]),
    "msl_iou_matrix": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "msl_multibox_match": (_I, [_P, _P, _P, _I, _P, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P]),
    "msl_multibox_match_count": (_I, [_P, _P, _P, _I, _P, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "msl_multibox_loss_pack_num_partials": (_I, [_I, _I]),
    "msl_multibox_loss_pack": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msl_multibox_loss_workspace_bytes": (_Z, []),
    "msl_multibox_loss_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "msl_multibox_loss_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "msl_multibox_loss_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "msl_multibox_loss_var_workspace_bytes": (_Z, [_I, _I]),
    "msl_multibox_loss_var": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "msl_detect_select_ws_ints": (_Z, [_I, _I, _I]),
    "msl_detect_objects": (_I, [_P, _P, _P, _I, _I, _I, _F, _F, _I] + [_P] * 15 + [_P]),
    "msl_adam_step": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "msl_nan_flag": (_I, [_P, _Z, _P, _I, _P]),
    "msl_nan_flag2": (_I, [_P, _Z, _I, _P, _Z, _I, _P, _P]),
    "msl_program_fn_id": (_I, [_P]),
    "msl_run_program": (_I, [_P, _P, _I, _I, _P]),
    "msl_run_program_mt": (_I, [_P, _P, _I, _I, _P, _P, _I, _P]),
    "msl_fill_u32": (_I, [_P, ctypes.c_uint, _Z, _P]),
    "msl_stem_dw_fwd_eval_supported": (_I, [_I] * 5),
    "msl_stem_dw_fwd_eval": (_I, [_P] * 6 + [_I] * 5 + [_P]),
    "msl_stem_dw_fwd_eval_bf16": (_I, [_P] * 6 + [_I] * 5 + [_P]),
    "msl_event_create": (_I, [_P]),
    "msl_event_create_device": (_I, [_P]),
    "msl_arm_stop_event": (_I, [_P, _I]),
    "msl_stop_event_pending": (_I, []),
    "msl_thread_launch_count": (_I, []),
    "msl_event_create_timed": (_I, [_P]),
    "msl_event_elapsed_ms": (_I, [_P, _P, _P]),
    "msl_event_destroy": (_I, [_P]),
    "msl_event_record": (_I, [_P, _P]),
    "msl_stream_wait_event": (_I, [_P, _P]),
}

_lib = None


class HipKernelError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipKernelError(
            f"{LIB_PATH} not found: build it with `make -C mslesions3d_amd/csrc` (or __graft_entry__.build()). "
            "mslesions3d_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_names():
    return sorted(_SIGNATURES.keys())


# True: the fork / join events inside a launch program are created with hipEventDisableSystemFence - they order streams of
# one GPU, where the kernel packets' own agent-scope fences already make the data visible; the system-scope release a default
# event adds at every record cost the recording stream 4-7 us each (15 records on the dependency chain of a training step:
# 0.718 -> 0.688 ms, same-box A/B).  Host-visible results still pass a system-scope release: the caller's stream waits for
# the chain through an ordinary torch event / stream synchronize.  tests/test_gpu_model.py::test_device_scope_events_*
DEVICE_SCOPE_EVENTS = True


def new_event(device_only=False):
    """A hipEvent (timing disabled) as an opaque integer handle.  ``device_only``: the event only ever orders streams of
    this GPU (fork / join inside a launch program) and may be created without the system-scope fence."""
    out = ctypes.c_void_p()
    if device_only and DEVICE_SCOPE_EVENTS:
        check(load().msl_event_create_device(ctypes.byref(out)), "msl_event_create_device")
    else:
        check(load().msl_event_create(ctypes.byref(out)), "msl_event_create")
    return out.value


# -- native replay ----------------------------------------------------------------------------------------------
ENQUEUE_THREADS = 2  # host threads issuing a replayed launch program (2: the side streams' calls go to a worker thread)
_SLOT_STRIDE = 28
_fn_ids = {}


def _slot(value, ctype):
    import struct
    if ctype is _F:
        return struct.unpack("<I", struct.pack("<f", value))[0]
    if ctype is _D:
        return struct.unpack("<Q", struct.pack("<d", value))[0]
    if value is None:
        return 0
    return int(value) & 0xFFFFFFFFFFFFFFFF


def _entry_stream(name, args):
    if name == "msl_event_record":
        return args[1]
    if name == "msl_stream_wait_event":
        return args[0]
    return args[-1]


STOP_EVENT_FORKS = True  # compile_program: launch + msl_event_record on the same stream -> launch carrying a stop event


def _fuse_stop_events(prog, timed_tags):
    """-> [(fn, args, tag, stream)]: an ``msl_event_record(ev, s)`` whose predecessor on stream ``s`` is an entry point
    that launched k >= 1 kernels becomes ``msl_arm_stop_event(ev, k - 1)`` in FRONT of that entry (the event then completes
    with the entry's last kernel: same meaning, no record packet on the stream - csrc/common.hpp, MSL_LAUNCH).  Only with
    launch counts from the recorder (Program.nl), never across a Python hook, never for timed launches."""
    nl = getattr(prog, "nl", None)
    fuse = STOP_EVENT_FORKS and nl is not None and len(nl) == len(prog)
    arm = getattr(load(), "msl_arm_stop_event")
    out, last = [], {}  # last[stream] = [entry, launches] of the newest entry on that stream if a record may fuse with it
    for k, (fn, args, tag) in enumerate(prog):
        if fn is None:
            out.append([None, (fn, args, tag, None)])
            last.clear()
            continue
        name = fn.__name__
        stream = _entry_stream(name, args)
        if name == "msl_event_record" and fuse and last.get(stream) is not None:
            item, launches = last.pop(stream)
            item[0] = (arm, (args[0], launches - 1), "stop_event", stream)
            continue
        item = [None, (fn, args, tag, stream)]
        out.append(item)
        launchy = name not in ("msl_event_record", "msl_stream_wait_event", "msl_arm_stop_event")
        last[stream] = [item, nl[k]] if (fuse and launchy and nl[k] >= 1 and tag not in timed_tags) else None
    flat = []
    for pre, ent in out:
        if pre is not None:
            flat.append(pre)
        flat.append(ent)
    return flat


def compile_program(prog, timed_tags=(), main_stream=None, device=0):
    """Recorded program -> list of segments: ("native", fn_ids, slots, n, tags, lanes, wait_for) runs in ONE foreign call
    through the generated trampolines (csrc/program_runner.hip); ("hook", callable) are the Python callbacks in between;
    launches whose tag is in ``timed_tags`` get an event record in front and behind (patched per replay).

    ``main_stream``: issue the segment with two host threads (msl_run_program_mt) - the calls on that stream stay with
    the caller, everything else goes to the worker thread; a stream-wait is held back until the event record it refers
    to has been issued.  None (or ``ENQUEUE_THREADS = 1``): one thread."""
    lib = load()
    two = main_stream is not None and ENQUEUE_THREADS != 1
    segs, ids, slots, tags, patches, lanes, waits = [], [], [], [], [], [], []
    last_record = {}

    def flush():
        if ids:
            n = len(ids)
            mt = None
            if two and any(lanes) and not all(lanes):
                mt = ((ctypes.c_int * n)(*lanes), (ctypes.c_int * n)(*waits), int(device))
            segs.append(("native", (ctypes.c_int * n)(*ids), (ctypes.c_ulonglong * (n * _SLOT_STRIDE))(*slots), n, list(tags), mt))
            ids.clear(), slots.clear(), tags.clear(), lanes.clear(), waits.clear()
            last_record.clear()  # records of an earlier segment have been issued by the time this one starts

    def push(fid, row, tag, lane, wait=-1):
        ids.append(fid)
        slots.extend(row + [0] * (_SLOT_STRIDE - len(row)))
        tags.append(tag)
        lanes.append(lane)
        waits.append(wait)

    armed = None
    for fn, args, tag, stream in _fuse_stop_events(prog, timed_tags):
        if fn is None:
            flush()
            segs.append(("hook", args))
            continue
        name = fn.__name__
        lane = 0 if (main_stream is None or stream == main_stream) else 1
        timed = tag in timed_tags
        if timed:  # event record / launch / event record back to back inside the native segment
            rec = _fn_ids.setdefault("msl_event_record", lib.msl_program_fn_id(b"msl_event_record"))
            patches.append((len(segs), len(ids) * _SLOT_STRIDE, (len(ids) + 2) * _SLOT_STRIDE, tag))
            push(rec, [0, _slot(stream, _P)], "event", lane)
        fid = _fn_ids.get(name)
        if fid is None:
            fid = _fn_ids[name] = lib.msl_program_fn_id(name.encode())
        if fid < 0:
            raise HipKernelError(f"{name} cannot be replayed natively")
        argtypes = _SIGNATURES[name][1]
        row = [_slot(a, t) for a, t in zip(args, argtypes)]
        assert len(row) == len(argtypes) <= _SLOT_STRIDE, name
        wait = -1
        if name == "msl_stream_wait_event":
            wait = last_record.get(args[1], -1)
        push(fid, row, tag, lane, wait)
        if name == "msl_event_record":
            last_record[args[0]] = len(ids) - 1
        elif name == "msl_arm_stop_event":
            armed = args[0]
        elif armed is not None:  # the entry the stop event rides on: a wait for the event may be issued once this one is
            last_record[armed] = len(ids) - 1
            armed = None
        if timed:
            push(rec, [0, _slot(stream, _P)], "event", lane)
    flush()
    return {"segments": segs, "patches": patches,
            "stop_event_forks": sum(1 for seg in segs if seg[0] == "native" for t in seg[4] if t == "stop_event")}


def new_timed_event():
    out = ctypes.c_void_p()
    check(load().msl_event_create_timed(ctypes.byref(out)), "msl_event_create_timed")
    return out.value


def destroy_event(ev):
    check(load().msl_event_destroy(ev), "msl_event_destroy")


def elapsed_ms(e0, e1):
    out = ctypes.c_float()
    check(load().msl_event_elapsed_ms(e0, e1, ctypes.byref(out)), "msl_event_elapsed_ms")
    return out.value


def replay_native(compiled, sink=None):
    """Run a compiled program.  If it has timed launches, a fresh timing-event pair is patched in around each of
    them (sink[tag] collects ("c", start, stop) handles; read them with elapsed_ms after a synchronise)."""
    lib = load()
    failed = ctypes.c_int(-1)
    segs = compiled["segments"]
    for seg_i, off0, off1, tag in compiled["patches"]:
        e0, e1 = new_timed_event(), new_timed_event()
        segs[seg_i][2][off0] = e0
        segs[seg_i][2][off1] = e1
        if sink is not None:
            sink.setdefault(tag, []).append(("c", e0, e1))
    for seg in segs:
        if seg[0] == "hook":
            seg[1]()
        else:
            if seg[5] is not None:
                rc = lib.msl_run_program_mt(seg[1], seg[2], _SLOT_STRIDE, seg[3], seg[5][0], seg[5][1], seg[5][2],
                                            ctypes.byref(failed))
            else:
                rc = lib.msl_run_program(seg[1], seg[2], _SLOT_STRIDE, seg[3], ctypes.byref(failed))
            if rc:
                check(rc, f"launch program entry {failed.value} ({seg[4][failed.value]})")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def check(code, what):
    if code != 0:
        raise HipKernelError(f"{what} failed with code {code} "
                             f"({'bad argument' if code == -1 else 'unsupported shape' if code == -2 else 'hipError'})")


# -- launch programs ------------------------------------------------------------------------------------------
# A training step is a fixed sequence of ~140 launches whose arguments (device pointers, sizes, stream) do not
# change from step to step.  While a recorder is active every `call` is also appended to it; `replay` then
# re-issues the sequence with no Python between two launches except the ctypes call itself.
_recorder = None


class Program(list):
    """A recorded launch program: (fn, args, tag) entries ((None, callable, tag) = Python hook) plus, per entry, the
    number of kernels the call launched (``nl``; compile_program needs it to turn a launch + event record into a launch
    with a stop event)."""

    def __init__(self, *a):
        super().__init__(*a)
        self.nl = []


def start_recording():
    global _recorder
    _recorder = Program()


def stop_recording():
    global _recorder
    prog, _recorder = _recorder, None
    return prog


def record_hook(fn, tag=None):
    """Insert a Python callback (e.g. 'start the all-reduce of bucket k') into the program being recorded."""
    if _recorder is not None:
        _recorder.append((None, fn, tag))
        _recorder.nl.append(0)


def call(name, *args, tag=None):
    """Invoke an int-returning entry point and raise on a non-zero code."""
    fn = getattr(load(), name)
    if _recorder is not None:
        count = load().msl_thread_launch_count
        n0 = count()
        rc = fn(*args)
        _recorder.append((fn, args, tag or name))
        _recorder.nl.append((count() - n0) & 0x7FFFFFFF)
        check(rc, name)
        return
    check(fn(*args), name)


def replay(prog, timed_tags=None, sink=None, event_factory=None):
    """Re-issue a recorded program.  ``timed_tags``/``sink``: record an event pair around the tagged launches."""
    for fn, args, tag in prog:
        if fn is None:
            args()
        elif timed_tags is not None and tag in timed_tags:
            e0, e1 = event_factory(), event_factory()
            e0.record()
            rc = fn(*args)
            e1.record()
            sink.setdefault(tag, []).append((e0, e1))
            if rc:
                check(rc, tag)
        else:
            rc = fn(*args)
            if rc:
                check(rc, tag)
