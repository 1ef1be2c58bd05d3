"""Regenerates the golden vectors under tests/golden/ from the CPU oracle (oracle/rt_oracle.c).

The reference ships no tests or golden images and cannot be executed here, so these
vectors pin the oracle against accidental change and give the GPU tests fixed
inputs/outputs; they are NOT outputs of the reference itself.

  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def main():
    s = O.OracleScene()
    s.default_scene()
    b = s.bindings()
    # (1) full frame 128x72, frames 0 and 1 accumulated, with the reference's debug counters
    W, H = 128, 72
    pc = O.default_pc(8)
    pb = O.pc_bytes(pc)
    O.camera_push_constants(pb, W=W, H=H)
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(2):
        pc["frame"] = f
        r = O.render(b, pb, W, H, accum=acc, counters=True)
        acc = r["accum"]
    np.savez_compressed(os.path.join(HERE, "default_128x72_f0f1.npz"), accum=r["accum"], out=r["out"],
                        tri=r["tri"].astype(np.int16), node=r["node"].astype(np.int16), rays=r["rays"].astype(np.int8))
    # (2) a 64x64 window of the 1280x720 configuration (over the dragon), frame 0
    W, H = 1280, 720
    pc = O.default_pc(8)
    pb = O.pc_bytes(pc)
    O.camera_push_constants(pb, W=W, H=H)
    win = (600, 300, 664, 364)
    r = O.render(b, pb, W, H, window=win, counters=True)
    x0, y0, x1, y1 = win
    np.savez_compressed(os.path.join(HERE, "default_1280x720_crop.npz"), window=np.array(win, np.int32),
                        accum=r["accum"][y0:y1, x0:x1], out=r["out"][y0:y1, x0:x1],
                        tri=r["tri"][y0:y1, x0:x1].astype(np.int16), node=r["node"][y0:y1, x0:x1].astype(np.int16))
    # (3) depth view (render mode 2) at 128x72
    W, H = 128, 72
    pc = O.default_pc(8)
    pc["renderMode"] = 2
    pb = O.pc_bytes(pc)
    O.camera_push_constants(pb, W=W, H=H)
    r = O.render(b, pb, W, H)
    np.savez_compressed(os.path.join(HERE, "default_128x72_depth.npz"), out=r["out"])
    # (4) per-corner vertex streams of the shipped models, as hashes (the OBJ reader pin that travels)
    import hashlib
    lines = []
    models = os.path.join(os.path.dirname(os.path.dirname(HERE)), "assets", "models")
    for f in sorted(os.listdir(models)):
        if f.endswith(".obj"):
            pos, nrm = O.obj_corners(os.path.join(models, f))
            lines.append("%s %d %s" % (f, pos.shape[0], hashlib.sha256(pos.tobytes() + nrm.tobytes()).hexdigest()))
    open(os.path.join(HERE, "obj_corner_hashes.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
