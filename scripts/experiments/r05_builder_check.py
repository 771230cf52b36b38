import sys, subprocess, tempfile
from pathlib import Path
ROOT = Path('/root/repo'); sys.path.insert(0, str(ROOT/'raytracing-one-weekend_amd'))
import numpy as np, torch, rtow
tmp = Path(tempfile.gettempdir())/'suz10.obj'
subprocess.run([sys.executable, str(ROOT/'scripts/make_mesh.py'), str(tmp), '10'], check=True, capture_output=True)
for name, scene in (('mesh100k', rtow.HostScene.obj(tmp, 16/9)), ('suzanne', rtow.HostScene.obj(ROOT/'tests/golden/suzanne.obj', 16/9))):
    for b in (rtow.BUILDER_DEVICE_LBVH, rtow.BUILDER_HOST_SAH):
        c = rtow.Context(0); c.set_builder(b)
        cfg = rtow.make_config(320, 180, 4, 2, 20, seed=1, precision=rtow.F64_FAST)
        import time
        c.render(scene, cfg)
        t=[]
        for _ in range(3):
            t0=time.perf_counter(); img, st = c.render(scene, cfg); t.append(time.perf_counter()-t0)
        bi = c.build_info()
        print(name, 'device' if b else 'host', 'bvh4_nodes', bi.bvh4_nodes, 'build_ms', round(bi.bvh_build_ms,3), 'node/seg', round(st.node_tests/st.segments,3), 'tri/seg', round(st.prim_tests/st.segments,3), 'call_ms', round(min(t)*1e3,2))
        c.close()
