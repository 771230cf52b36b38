"""Regenerates the data fixtures under tests/golden/ (run in the build container,
where /root/reference exists; the GPU box only sees the committed outputs).

  suzanne.obj        the reference's only mesh asset (its suzanne.obj, 511 v / 968 f),
                     reduced to the `v` and `f` records the loader reads (data, not code)
  oracle_philox.npz  small images rendered by oracle/ with the Philox stream, so the
                     GPU box can check the device path even if it rebuilt nothing
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import orc  # noqa: E402
import rtow  # noqa: E402

REF_OBJ = Path("/root/reference/suzanne.obj")


def make_suzanne():
    out = []
    for line in REF_OBJ.read_text().splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            out.append("v " + " ".join(t[1:4]))
        elif t[0] == "f":
            out.append("f " + " ".join(x.split("/")[0] for x in t[1:]))
    (HERE / "suzanne.obj").write_text("\n".join(out) + "\n")
    print("suzanne.obj:", sum(l[0] == "v" for l in out), "v", sum(l[0] == "f" for l in out), "f")


CASES = {
    # name: (scene ctor, width, aspect, spp, nstreams, depth, seed)
    "c1_3spheres": (lambda: orc.OrcScene.cover(0, 16 / 9, True), 64, 16 / 9, 8, 2, 10, 7),
    "cover_static": (lambda: orc.OrcScene.cover(11, 1.5, False), 60, 1.5, 4, 2, 50, 1),
    "cover_moving": (lambda: orc.OrcScene.cover(11, 1.5, True), 60, 1.5, 4, 1, 50, 3),
    "suzanne": (lambda: orc.OrcScene.obj(HERE / "suzanne.obj", 16 / 9), 64, 16 / 9, 4, 2, 20, 5),
}


def make_philox_goldens():
    data = {}
    for name, (ctor, w, aspect, spp, ns, depth, seed) in CASES.items():
        sc = ctor()
        cfg = rtow.make_config(w, rtow.image_height(w, aspect), spp, ns, depth, seed=seed)
        img, st = orc.render(sc, cfg, orc.RNG_PHILOX, nthreads=4)
        data[name + "_img"] = img
        data[name + "_segments"] = np.array([st.segments], dtype=np.uint64)
        print(name, img.shape, "segments", st.segments)
    np.savez_compressed(HERE / "oracle_philox.npz", **data)


if __name__ == "__main__":
    make_suzanne()
    make_philox_goldens()
