"""Dump the recorded launch program of one training step (GPU box): index, lane (0 = chain), entry point, tag, and for
stream waits the index of the record they depend on.  Usage: python tools/probes/program_dump.py [f32|bf16] > out.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd.ssd3d import LSSD3D, MultiBoxLoss  # noqa: E402
from mslesions3d_amd.synth import make_batch_on_device  # noqa: E402
from mslesions3d_amd.trainer import FusedTrainer  # noqa: E402

dev = torch.device("cuda", 0)


def dump_train(dtype):
    """The training step's program (FusedTrainer, 128^3 x 4)."""
    torch.manual_seed(0)
    model = LSSD3D(n_classes=2, input_channels=1, input_size=(128,) * 3, threshold=[0.1, 0.2], lr=1e-3).to(dev).train()
    model.compute_dtype = dtype
    tr = FusedTrainer(model)
    x, b, l = make_batch_on_device(4, (128,) * 3, dev, 1, seed=1)
    packed = (x,) + MultiBoxLoss.pack_targets(b, l, dev)
    for _ in range(3):
        tr.step_packed(*packed, sync=False, resident=True)
    torch.cuda.synchronize()
    entry = list(tr._programs.values())[-1]
    main = tr._stream.cuda_stream
    flat = _lib._fuse_stop_events(entry["prog"], ())
    last_record, armed = {}, None
    for i, (fn, args, tag, stream) in enumerate(flat):
        if fn is None:
            print(f"{i:4d}  hook {tag}")
            continue
        name = fn.__name__
        lane = 0 if stream == main else 1
        dep = ""
        if name == "msl_stream_wait_event":
            dep = f"  <- record at {last_record.get(args[1], '?')}"
        print(f"{i:4d}  lane {lane}  {name:44s} {tag}{dep}")
        if name == "msl_event_record":
            last_record[args[0]] = i
        elif name == "msl_arm_stop_event":
            armed = args[0]
        elif armed is not None:
            last_record[armed] = i
            armed = None


def dump_predict(dtype):
    """The same for LSSD3D.predict_step's launch program at 192^3 x 2 (`python tools/probes/program_dump.py predict [bf16]`)."""
    m = LSSD3D(n_classes=2, input_channels=1, input_size=(192,) * 3, threshold=[0.1, 0.2]).to(dev).eval()
    m.compute_dtype = dtype
    xs = make_batch_on_device(2, (192,) * 3, dev, 1, seed=1)[0]
    for _ in range(2):
        m.predict_step({"img": xs})
    ent = list(m._pred_programs.values())[-1]
    main_s = torch.cuda.current_stream().cuda_stream
    last, armed_ = {}, None
    for i, (fn, args, tag, stream) in enumerate(_lib._fuse_stop_events(ent["prog"], ())):
        name = fn.__name__
        dep = f"  <- record at {last.get(args[1], '?')}" if name == "msl_stream_wait_event" else ""
        print(f"{i:4d}  lane {0 if stream == main_s else 1}  {name:44s} {tag}{dep}")
        if name == "msl_event_record":
            last[args[0]] = i
        elif name == "msl_arm_stop_event":
            armed_ = args[0]
        elif armed_ is not None:
            last[armed_] = i
            armed_ = None


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "predict":
        dump_predict(sys.argv[2] if len(sys.argv) > 2 else "f32")
    else:
        dump_train(sys.argv[1] if len(sys.argv) > 1 else "f32")
