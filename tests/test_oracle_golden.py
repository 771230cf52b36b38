"""The oracle against the reference's outputs: the five PPM md5 sums recorded in
SURVEY.md §8c (reference at -t 1, default mt19937 seed), plus the BVH diagnostics the
reference prints.  This is what pins oracle/ to the reference."""
import hashlib
import json

import pytest

import orc
import rtow
from conftest import GOLDEN

CASES = json.loads((GOLDEN / "survey_md5.json").read_text())["cases"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_ppm_md5_matches_reference(case):
    if case["scene"] == "cover":
        scene = orc.OrcScene.cover(case["nsqrt"], case["aspect"], case["moving"])
    else:
        scene = orc.OrcScene.obj(GOLDEN / "suzanne.obj", case["aspect"])
    assert scene.c.n_prims == case["prims"]
    w = case["width"]
    h = rtow.image_height(w, case["aspect"])
    cfg = rtow.make_config(w, h, case["spp"], 1, case["depth"])
    img, st = orc.render(scene, cfg, orc.RNG_MT19937)
    txt = orc.ppm_text(img, w, h, case["spp"])
    assert hashlib.md5(txt).hexdigest() == case["md5"]
    if "first_pixel" in case:
        assert txt.split(b"\n")[3].decode() == case["first_pixel"]
    if "stupid_volume" in case:  # "Total BVH stupid volume" (src/render.cpp:148), 6 sig. digits
        assert float("%.6g" % st.bvh_stupid_volume) == case["stupid_volume"]


def test_survey_workload_statistics():
    """Per-sample work the survey measured on the reference binary."""
    scene = orc.OrcScene.cover(11, 1.5, False)
    cfg = rtow.make_config(300, 200, 8, 1, 50)
    _, st = orc.render(scene, cfg, orc.RNG_MT19937)
    assert abs(st.segments / st.samples - 2.43) < 0.01
    assert abs(st.node_tests / st.segments - 58.9) < 0.1
    assert abs(st.prim_tests / st.segments - 65.9) < 0.1
    assert abs(st.rng_doubles / st.samples - 13.9) < 0.05
    assert (st.bvh_nodes, st.bvh_leaves) == (255, 128)
