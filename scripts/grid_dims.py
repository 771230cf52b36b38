#!/usr/bin/env python3
"""Grid resolution and image size of the cover scenes (diagnostic)."""
import struct, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
import rtow
for moving in (False, True):
    ctx = rtow.Context(0)
    ctx.upload(rtow.HostScene.cover(11, 1.5, moving))
    img = ctx.debug_image(1)
    n = struct.unpack_from("<3i", img, 36)
    cell = struct.unpack_from("<3f", img, 12)
    print("moving" if moving else "static", "cells", n, "cell size", [round(c, 4) for c in cell], "image bytes", len(img),
          "fat stride", struct.unpack_from("<I", img, 60)[0])
