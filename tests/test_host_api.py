"""Host side of the product (no GPU): the C-ABI library loads and exports what
include/rtow.h declares, the scene scripts flatten to exactly the oracle's scenes,
the PPM writer matches, row partitioning, argument validation, the CLI."""
import ctypes as C
import re
import subprocess
import sys

import numpy as np
import pytest

import orc
import rtow
from conftest import GOLDEN, REPO


def test_library_exports_every_declared_symbol():
    header = (REPO / "include" / "rtow.h").read_text()
    declared = set(re.findall(r"\b(rtow_[a-z0-9_]+)\s*\(", header))
    declared -= {"rtow_render_device_"}
    assert declared, "no declarations parsed"
    L = rtow.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in rtow.h but not exported"
    assert set(rtow.EXPORTS) <= declared
    assert L.rtow_abi_version() == rtow.RTOW_ABI_VERSION


@pytest.mark.parametrize("moving", [False, True])
def test_cover_scene_flattens_like_the_oracle(moving):
    mine = rtow.HostScene.cover(11, 1.5, moving)
    ref = orc.OrcScene.cover(11, 1.5, moving)
    a, b = orc.scene_arrays(mine.c), orc.scene_arrays(ref.c)
    assert mine.c.n_prims == (485 if moving else 486)  # default-seed counts
    for k in b:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("moving", [False, True])
def test_variant_store_and_world_flatten_like_the_oo_scene(moving):
    """the reference's other two scene models — VariantStore primitives
    (src/variant-primitives.h:84-113) and the World of src/vmodel.h:250-253 — built by the same calls of
    lots_of_balls() flatten to the rtow_scene_t of the OO scene, record for record."""
    oo_scene = rtow.HostScene.cover(11, 1.5, moving)  # (kept alive: scene_arrays are views)
    oo = orc.scene_arrays(oo_scene.c)
    for model in (rtow.MODEL_VARIANT, rtow.MODEL_WORLD):
        other = rtow.HostScene.cover(11, 1.5, moving, model=model)
        assert other.c.n_prims == (485 if moving else 486)
        b = orc.scene_arrays(other.c)
        assert set(b) == set(oo)
        for k in oo:
            assert np.array_equal(oo[k], b[k]), (model, k)


def test_variant_store_obj_scene_and_world_refusal():
    oo_scene = rtow.HostScene.obj(GOLDEN / "suzanne.obj")
    oo = orc.scene_arrays(oo_scene.c)
    var = rtow.HostScene.obj(GOLDEN / "suzanne.obj", model=rtow.MODEL_VARIANT)
    assert var.c.n_triangles == 968
    b = orc.scene_arrays(var.c)
    for k in oo:
        assert np.array_equal(oo[k], b[k]), k
    with pytest.raises(rtow.RtowError):  # src/vmodel.h's World = VariantStore<Sphere, MovingSphere>: no triangles
        rtow.HostScene.obj(GOLDEN / "suzanne.obj", model=rtow.MODEL_WORLD)


def test_obj_scene_flattens_like_the_oracle():
    mine = rtow.HostScene.obj(GOLDEN / "suzanne.obj")
    ref = orc.OrcScene.obj(GOLDEN / "suzanne.obj")
    a, b = orc.scene_arrays(mine.c), orc.scene_arrays(ref.c)
    assert mine.c.n_triangles == 968
    for k in b:
        assert np.array_equal(a[k], b[k]), k


def test_obj_errors(tmp_path):
    quad = tmp_path / "line.obj"  # (a quad is not an error: tinyobj triangulates it, src/main.cpp:109)
    quad.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2\n")
    out = C.POINTER(rtow.Scene)()
    hc = rtow.HostConfig(0, 1.5, 0)
    L = rtow.lib()
    # "Oops found a face that isn't a triangle" (src/main.cpp:130) -> error code, no abort
    assert L.rtow_host_scene_obj(C.byref(hc), str(quad).encode(), C.byref(out)) == rtow.RTOW_EINVAL
    assert L.rtow_host_scene_obj(C.byref(hc), b"/nonexistent.obj", C.byref(out)) == rtow.RTOW_EINVAL


def test_ppm_writer_matches_oracle_writer():
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 40, size=(7, 9, 3))
    img[0, 0] = [0.0, 1e9, 16 * 0.999**2]  # clamp low, clamp high, edge
    a = rtow.ppm_text(img, 9, 7, 16)
    b = orc.ppm_text(img, 9, 7, 16)
    assert a == b and a.startswith(b"P3\n9 7\n255\n0 255 255\n")


def test_local_rows_partition():
    H = 37
    seen = []
    for r in range(3):
        cfg = rtow.make_config(5, H, 1, rank=r, nranks=3, tile_rows=4)
        rows = rtow.local_rows(cfg)
        assert rows == [i for i in range(H) if (i // 4) % 3 == r]
        seen += rows
    assert sorted(seen) == list(range(H))


def test_config_validation_without_gpu():
    L = rtow.lib()
    bad = rtow.make_config(0, 10, 1)
    assert L.rtow_local_rows(C.byref(bad)) == rtow.RTOW_EINVAL
    assert b"image size" in L.rtow_last_error()
    bad = rtow.make_config(10, 10, 1, nstreams=0)
    assert L.rtow_local_rows(C.byref(bad)) == rtow.RTOW_EINVAL
    bad = rtow.make_config(10, 10, 1, rank=2, nranks=2)
    assert L.rtow_local_rows(C.byref(bad)) == rtow.RTOW_EINVAL


def test_cli_dry_run_prints_the_reference_config_text():
    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    out = subprocess.run([str(exe), "--dry-run", "-w", "400", "-s", "10", "-t", "2", "-a", "2"],
                         capture_output=True, text=True, check=True).stdout
    assert out == ("Config {\naspect_ratio: 2\nnumber_of_balls_sqrt: 11\nmoving_spheres: 1\n"
                   "image_width: 400\nsamples_per_pixel: 10\nmax_child_rays: 20\nnthreads: 2\n}\n")
    r = subprocess.run([str(exe), "--bogus"], capture_output=True, text=True)
    assert r.returncode != 0 and "not expected" in r.stderr


def test_rtweekend_names_the_frames_of_a_fatal_signal_and_keeps_the_signal_as_exit_status():
    """Round 4 lost one `rtweekend` run to a SIGSEGV after the image was complete and had no frame to show for it; the
    answer then was `_Exit(0)`.  Now main returns like the reference's (src/main.cpp:165-170) and a fatal signal writes
    the phase and a raw backtrace to stderr before it is re-raised with the default action: the exit status is still
    the signal's (no masking), and the next occurrence in any ordinary run names where it was."""
    import signal

    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    src = (REPO / "raytracing-one-weekend_amd" / "host" / "main.cpp").read_text()
    assert "_Exit" not in src and "quick_exit" not in src
    r = subprocess.run([str(exe), "-w", "16", "-s", "4", "-t", "1", "-n", "0"], capture_output=True, text=True,
                       env=dict(__import__("os").environ, RTOW_TEST_RAISE=str(int(signal.SIGSEGV))))
    assert r.returncode == -signal.SIGSEGV, r.returncode
    assert "fatal signal 11" in r.stderr and "phase: parsing flags" in r.stderr
    assert "backtrace" in r.stderr and "rtweekend" in r.stderr.split("backtrace", 1)[1]  # at least its own frame
    assert not r.stdout


def _check_compact_line(line, raw_len):
    """The contract of the ONE line bench.py prints (benchline.py): under the cap, exactly the agreed keys."""
    import json

    import benchline

    base = json.loads((REPO / "BASELINE.json").read_text())
    assert raw_len < benchline.LINE_CAP
    for key in benchline.CONTRACT_KEYS:
        assert key in line, key
    assert set(line) <= set(benchline.CONTRACT_KEYS) | {"value_e2e", "scaling_base_value", "projected_efficiency_n8",
                                                        "projected_efficiency_n8_with_handle", "multi_handle", "other",
                                                        "details"}
    assert set(line["config"]) == set(benchline.CONFIG_KEYS)
    assert set(line["roofline"]) == set(benchline.ROOFLINE_KEYS) | {"hbm_equivalent_GBps"}
    assert line["metric"] == base["metric"] and line["unit"] == "Msamples/s"
    assert line["vs_baseline"] is None and line["higher_is_better"] is True
    assert not any(isinstance(v, (dict, list)) for v in (line["other"] or {}).values())  # plain numbers
    assert all(len(v) <= 240 for v in _strings(line)), "prose belongs in DESIGN.md, not in the line"


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)


def test_compact_bench_line_from_committed_details_stays_under_the_cap():
    """BENCH_r04.json had `parsed: null`: the one line had grown to 20.9 KB.  bench.py now prints
    benchline.compact_line(details) and writes the details to a file.  Built here from the committed details of the
    driver's command (rounds 4 and 5): under 4 KB, the contract's keys, roofline and cpu_baseline as objects,
    everything else scalars."""
    import json

    import benchline

    for name in ("r04_bench_driver_style.json", "r05_bench_details.json"):
        f = REPO / "profiles" / name
        if not f.exists():
            continue
        details = json.loads(f.read_text())
        line = benchline.compact_line(details, "gpurun_out/bench_details.json")
        raw = benchline.dumps(line)
        assert "\n" not in raw
        _check_compact_line(json.loads(raw), len(raw.encode()))
        assert line["value"] == details["value"] and line["roofline"]["frac"] == details["roofline"]["frac"]
        assert line["cpu_baseline"]["kind"] in ("port", "reference") and line["cpu_baseline"]["cores"] >= 1
        assert line["other"]["mesh100k"] > 0 and line["other"]["stream_stress"] > 0
    # a line that outgrows the cap is an error, not a silently unparsed record
    fat = benchline.compact_line(details, "x" * 5000)
    with pytest.raises(ValueError):
        benchline.dumps(fat)


def test_committed_bench_line_follows_the_contract():
    """The compact line committed under profiles/ (the driver's command, final sources of the round) has every
    field the driver's contract names, is consistent with BASELINE.json, and carries roofline and cpu_baseline."""
    import json

    f = REPO / "profiles" / "r05_bench_line.json"
    if not f.exists():
        pytest.skip("no round-5 line committed yet")
    raw = f.read_text().strip()
    assert raw.count("\n") == 0
    line = json.loads(raw)
    _check_compact_line(line, len(raw.encode()))
    assert line["dtype"] == "f64" and line["n_gpus"] == 1
    assert "1200x800, 100 spp, 50 bounces" in line["config"]["workload"] and line["config"]["baseline_config"] == "configs[1]"
    r = line["roofline"]
    assert r["bound"] == "valu_issue" and r["peak"] == 1.0 and r["kernel_ms"] <= line["ms_per_step"]
    if r["frac"] is not None:
        assert abs(r["frac"] - min(r["issue_utilisation"], 1.0) * r["lane_activity"]) < 1e-3
    samples = 1200 * 800 * line["config"]["spp_effective"]
    assert abs(line["value"] - samples / (line["ms_per_step"] * 1e-3) / 1e6) / line["value"] < 1e-3
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "Msamples/s"


def test_general_obj_loads_all_shapes_and_triangulates(tmp_path, monkeypatch):
    """The reference reads shapes[0] only (src/main.cpp:115) and gets polygons fan-triangulated by
    tinyobj 1.0.6's default triangulate = true (src/main.cpp:109), which the default path restates.
    Beyond the reference: RTOW_GENERAL_OBJ=1 / --general-obj loads every shape; relative (negative)
    indices resolve against the vertices read so far.  A face of two vertices is the reference's
    "isn't a triangle" error (:130)."""
    obj = tmp_path / "two_shapes.obj"
    obj.write_text("o first\nv -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nf 1 2 3 4\n"
                   "o second\nv -1 -1 -1\nv 1 -1 -1\nv 0 1 -1\nf -3 -2 -1\n")
    first = rtow.HostScene.obj(obj)  # reference behaviour: first shape only, the quad as a fan
    assert first.c.n_triangles == 2
    t2 = np.ctypeslib.as_array(first.c.triangle_geom, shape=(2, 9))
    assert np.array_equal(t2[0], [-1, -1, 0, 1, -1, 0, 1, 1, 0]) and np.array_equal(t2[1], [-1, -1, 0, 1, 1, 0, -1, 1, 0])
    bad = tmp_path / "line.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nf 1 2\n")
    with pytest.raises(Exception):
        rtow.HostScene.obj(bad)
    monkeypatch.setenv("RTOW_GENERAL_OBJ", "1")
    sc = rtow.HostScene.obj(obj)
    assert sc.c.n_triangles == 3
    tri = np.ctypeslib.as_array(sc.c.triangle_geom, shape=(3, 9))
    assert np.array_equal(tri[0], [-1, -1, 0, 1, -1, 0, 1, 1, 0])      # fan: (v0, v1, v2)
    assert np.array_equal(tri[1], [-1, -1, 0, 1, 1, 0, -1, 1, 0])      #      (v0, v2, v3)
    assert np.array_equal(tri[2], [-1, -1, -1, 1, -1, -1, 0, 1, -1])   # second shape, relative indices


def test_product_path_fails_loudly_without_a_hip_device():
    """No CPU fallback: on a machine without a GPU every device entry point reports RTOW_ENODEV /
    RTOW_EHIP instead of rendering on the host (the oracle is test infrastructure only)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(rtow.RtowError) as ei:
        rtow.Context(0)
    assert "HIP" in str(ei.value) or "device" in str(ei.value)
    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    r = subprocess.run([str(exe), "-w", "16", "-s", "4", "-t", "1", "-n", "0"], capture_output=True)
    assert r.returncode != 0 and not r.stdout.startswith(b"P3")
    # bench.py refuses too
    b = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True)
    assert b.returncode != 0 and b"no CPU fallback" in b.stderr + b.stdout
    # ... and `--gpus N` without N devices exits non-zero before starting any rank (never a silent 1-GPU run)
    for extra in (["--gpus", "8"], ["--gpus", "2", "--backend", "gloo"]):
        b = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "1", "--warmup", "0"] + extra,
                           capture_output=True)
        assert b.returncode != 0 and b"HIP device" in b.stderr and b"n_gpus" not in b.stdout


def _schedule(cfg):
    pairs = (C.c_uint32 * 4096)()
    n = rtow.lib().rtow_debug_schedule(None, C.byref(cfg), pairs, 2048)
    assert n >= 0
    return [(pairs[2 * i], pairs[2 * i + 1]) for i in range(n)]


def _expected_levels(s0, total, chunk=10):
    """level_plan of csrc/rtow_capi.cpp restated: the divisor of gcd(first sample, sample count) nearest `chunk` in
    log distance (ties: the smaller) when it lies within a factor of two of it; otherwise levels of exactly `chunk`
    samples with the remainder added to the last one (a range shorter than a chunk: one level)."""
    import math
    g = math.gcd(total, s0)
    best = min((e for e in range(1, g + 1) if g % e == 0), key=lambda e: (round(abs(math.log(e / chunk)), 12), e))
    if chunk <= 2 * best <= 4 * chunk:
        return [(s0 + k * best, best) for k in range(total // best)]
    n = max(total // chunk, 1)
    d = chunk if n > 1 else total
    return [(s0 + k * d, d) for k in range(n - 1)] + [(s0 + (n - 1) * d, total - (n - 1) * d)]


@pytest.mark.parametrize("spp,ns", [(100, 10), (100, 4), (500, 4), (99, 3), (7, 3), (1, 1), (16, 1), (1024, 64), (97, 1),
                                    (101, 1), (127, 1), (131, 1), (202, 2), (23, 1), (1009, 1)])
def test_fast_build_schedule_properties(spp, ns, monkeypatch):
    """The levels a fast-build render is cut into (pure host arithmetic): runs that tile the effective sample range
    — the divisor of the range nearest RTOW_SCHED_CHUNK (10) when there is one within a factor of two, else runs of
    10 with the remainder on the last — whatever nstreams is; the strict build keeps one level per stream, the
    reference's decomposition (src/render.cpp:169-185).  Never `spp` levels of one sample (101, 127, 131: primes)."""
    monkeypatch.delenv("RTOW_SCHED_CHUNK", raising=False)
    eff = spp // ns * ns
    s = _schedule(rtow.make_config(64, 48, spp, ns, 10, precision=rtow.F64_FAST))
    assert s[0][0] == 0 and sum(c for _, c in s) == eff
    assert all(a + c == b for (a, c), (b, _) in zip(s, s[1:]))
    assert s == _expected_levels(0, eff)
    d = s[0][1]
    assert all(c == d for _, c in s[:-1]) and d <= s[-1][1] < 2 * max(d, 10)
    if eff >= 10:  # bounded work items and partial images: nothing shorter than half a chunk, nothing longer than two
        assert all(5 <= c <= 20 for _, c in s), s
    # the same effective spp through another stream count: the same levels
    assert _schedule(rtow.make_config(64, 48, eff, 1, 10, precision=rtow.F64_FAST)) == s
    # strict build: one level per stream
    strict = rtow.make_config(64, 48, spp, ns, 10, precision=rtow.F64_STRICT)
    assert _schedule(strict) == [(k * (spp // ns), spp // ns) for k in range(ns)]
    # a stream range covers its own samples
    if ns >= 2:
        part = rtow.make_config(64, 48, spp, ns, 10, precision=rtow.F64_FAST, stream_first=1, stream_count=ns - 1)
        ps = _schedule(part)
        assert ps[0][0] == spp // ns and sum(c for _, c in ps) == (ns - 1) * (spp // ns)
        assert ps == _expected_levels(spp // ns, (ns - 1) * (spp // ns))
    monkeypatch.setenv("RTOW_SCHED_CHUNK", "0")  # off: the strict build's levels
    assert _schedule(rtow.make_config(64, 48, spp, ns, 10, precision=rtow.F64_FAST)) == _schedule(strict)


@pytest.mark.parametrize("which", ["cover", "moving", "suzanne", "tiny"])
def test_product_reference_tree_builder_matches_the_oracles_tree(which):
    """csrc/rtow_reftree.h (the product's restatement of the reference's BVH build, src/render.cpp:73-110, which
    RTOW_KERNEL_REFTREE walks on the device) against the oracle's restatement of the same build, without a GPU:
    node count and the reference's "Total BVH stupid volume" diagnostic — a sum over every node's box, so it
    sees the boxes (float-rounded triangle corners, origin-unioned leaves), the split axes and the sort order —
    and against the values SURVEY.md §8c records from the reference itself (2150.93 / 34.6011)."""
    if which == "suzanne":
        scene = rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9)
        oscene, cfg, survey = orc.OrcScene.obj(GOLDEN / "suzanne.obj", 16 / 9), rtow.make_config(8, 4, 1, 1, 2), 34.6011
    else:
        n, moving = (0, False) if which == "tiny" else (11, which == "moving")
        scene = rtow.HostScene.cover(n, 1.5, moving)
        oscene, cfg = orc.OrcScene.cover(n, 1.5, moving), rtow.make_config(8, 4, 1, 1, 2)
        survey = 2150.93 if which == "cover" else None
    nodes, depth, vol = C.c_int32(), C.c_int32(), C.c_double()
    assert rtow.lib().rtow_host_reftree_info(C.byref(scene.c), C.byref(nodes), C.byref(depth), C.byref(vol)) == 0
    _, ost = orc.render(oscene, cfg, orc.RNG_MT19937, nthreads=1)  # (the oracle builds the reference's tree per render)
    assert nodes.value == ost.bvh_nodes
    assert vol.value == ost.bvh_stupid_volume  # the same operations in the same order: equal, not close
    if survey is not None:
        assert float("%.6g" % vol.value) == survey
    assert 0 <= depth.value < 48


def test_round4_bench_line_carries_the_blocks_the_review_asked_for():
    """profiles/r04_bench_driver_style.json (the driver's command on the final sources of round 4): the contract's
    fields, the roofline object as the BINDING bound with the metric's equivalent-streaming figure beside it, the
    per-rank projection of configs[2] at N = 2 / 4 / 8, the multi-device handle's per-frame cost, and what a caller
    waits for on the meshes with either builder."""
    import json

    base = json.loads((REPO / "BASELINE.json").read_text())
    line = json.loads((REPO / "profiles" / "r04_bench_driver_style.json").read_text())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "scale_projection", "multi_handle",
                "value_e2e", "scaling_base", "other_configs"):
        assert key in line, key
    assert line["metric"] == base["metric"] and line["dtype"] == "f64" and line["n_gpus"] == 1 and line["vs_baseline"] is None
    assert "1200x800, 100 spp, 50 bounces" in line["config"]["workload"]
    samples = 1200 * 800 * line["config"]["spp_effective"]
    assert abs(line["value"] - samples / (line["ms_per_step"] * 1e-3) / 1e6) / line["value"] < 1e-3
    r = line["roofline"]
    assert r["bound"] == "valu_issue" and r["pmc_stale"] is False and r["kernel_ms"] <= line["ms_per_step"]
    assert abs(r["frac"] - min(r["issue_utilisation"], 1.0) * r["lane_activity"]) < 1e-3 and 0.3 < r["frac"] < 0.7
    assert r["traffic"] > 2e8  # HBM bytes per launch (PMC): the partial sums
    h = r["hbm_equivalent_streaming"]
    assert h["bound"] == "hbm" and h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-4
    sp = line["scale_projection"]["by_n"]
    assert [b["n_gpus"] for b in sp] == [2, 4, 8] and [b["tile_rows"] for b in sp] == [8, 8, 4]
    for b in sp:
        assert len(b["per_rank"]) == b["n_gpus"] and b["kernel_ms_max"] >= b["kernel_ms_mean"] > 0
        assert abs(b["projected_Msamples_per_s"] - 1200 * 800 * 500 / (b["step_ms_max"] * 1e-3) / 1e6) / b["projected_Msamples_per_s"] < 1e-3
        assert 0.5 < b["projected_efficiency"] <= 1.02
    m = line["multi_handle"]
    assert m["use_rccl"] is True and 0.9 < m["rgb8"]["over_rtow_render_rgb8"] < 1.1
    meshes = [o for o in line["other_configs"] if "end_to_end_rgb8" in o]
    assert len(meshes) == 2 and all({"host_sah", "device_lbvh"} <= set(o["end_to_end_rgb8"]) for o in meshes)
    assert all(o["end_to_end_rgb8"]["device_lbvh"]["bvh4_nodes"] > 0 for o in meshes)  # the device builder makes the 4-wide image
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "Msamples/s"
