"""benchline.py — the ONE line bench.py prints, cut from the full record of a run.

bench.py measures a lot (per-rank tables, per-leg rooflines, the builders' end-to-end legs, the multi-device
handle).  All of that goes to a details file (`--details-out`) and to stderr; stdout carries one compact JSON
object, at most LINE_CAP bytes, with the driver's contract keys, `roofline`, `cpu_baseline` and a handful of
scalars.  Pure dictionary work: no torch, no HIP, importable on a machine without a GPU (the CPU test builds
the line from a committed details file and checks the cap and the key set).

The reference's own timing facility is one stderr line (`src/render.cpp:188-190`); the prose that explains the
numbers lives in DESIGN.md §5, not in the JSON.
"""
from __future__ import annotations

import json

LINE_CAP = 4096  # bytes; the round-4 line (20.9 KB) was not parsed by the driver, the round-3 one (13.3 KB) was

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
CONFIG_KEYS = ("workload", "baseline_config", "spp_effective", "precision", "kernel", "segments_per_sample")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "issue_utilisation",
                 "lane_activity", "pmc_file", "pmc_stale", "hbm_equivalent_frac", "measured_hbm_GBps")


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _short_kernel(name):
    """'grid (per-lane 3D-DDA …)' -> 'grid'"""
    return name.split(" ", 1)[0] if isinstance(name, str) else name


def _other(details):
    """The other BASELINE configs as plain numbers (Msamples/s)."""
    out = {}
    for o in details.get("other_configs") or []:
        base = o.get("baseline_config", "")
        if "streaming stress" in base:
            out["stream_stress"] = o.get("value")
            out["stream_stress_GBps"] = _get(o, "roofline", "hbm_equivalent_streaming", "achieved")
        elif "moving" in base:
            out["moving"] = o.get("value")
        elif base.startswith("configs[3]"):
            out["suzanne"] = o.get("value")
            out["suzanne_e2e"] = _get(o, "end_to_end_rgb8", "value_e2e")
        elif base.startswith("configs[4]"):
            out["mesh100k"] = o.get("value")
            out["mesh100k_e2e"] = _get(o, "end_to_end_rgb8", "value_e2e")
            out["mesh100k_frac"] = _get(o, "roofline", "frac")
            out["mesh100k_e2e_32spp_auto"] = _get(o, "end_to_end_rgb8_32spp", "auto", "value")
            out["mesh100k_e2e_32spp_host"] = _get(o, "end_to_end_rgb8_32spp", "host_sah", "value")
    return out or None


def compact_line(details, details_path=None):
    """The compact object of one bench run (dict in, dict out)."""
    line = {k: details.get(k) for k in CONTRACT_KEYS if k not in ("config", "roofline", "cpu_baseline")}
    cfg = details.get("config") or {}
    line["config"] = {k: cfg.get(k) for k in CONFIG_KEYS}
    line["config"]["kernel"] = _short_kernel(line["config"]["kernel"])
    r = details.get("roofline") or {}
    roof = {k: r.get(k) for k in ROOFLINE_KEYS if k != "hbm_equivalent_frac"}
    roof["unit"] = "active VALU lane-slot fraction" if r.get("bound") == "valu_issue" else r.get("unit")
    roof["hbm_equivalent_frac"] = _get(r, "hbm_equivalent_streaming", "frac")
    roof["hbm_equivalent_GBps"] = _get(r, "hbm_equivalent_streaming", "achieved")
    line["roofline"] = roof
    c = details.get("cpu_baseline")
    if c:
        c = dict(c)
        if len(c.get("sample", "")) > 200:
            c["sample"] = c["sample"][:200]
    line["cpu_baseline"] = c
    line["value_e2e"] = details.get("value_e2e")
    line["scaling_base_value"] = _get(details, "scaling_base", "value")
    by_n = _get(details, "scale_projection", "by_n") or []
    for b in by_n:
        if b.get("n_gpus") == 8:
            line["projected_efficiency_n8"] = b.get("projected_efficiency")
            line["projected_efficiency_n8_with_handle"] = b.get("projected_efficiency_with_handle")
    mh = details.get("multi_handle") or {}
    if "rgb8" in mh:
        line["multi_handle"] = {"ms_per_frame_rgb8": _get(mh, "rgb8", "ms_per_frame"),
                                "over_kernel_ms": _get(mh, "rgb8", "over_kernel_ms"),
                                "over_rtow_render_rgb8": _get(mh, "rgb8", "over_rtow_render_rgb8")}
    line["other"] = _other(details)
    fif = _get(details, "two_frames_in_flight", "value")
    if fif is not None and line["other"] is not None:
        line["other"]["cover_two_frames_in_flight"] = fif  # (a sequence of frames, two in flight: never `value`)
    line["details"] = details_path
    return line


def dumps(line):
    """One line of JSON, ASCII-safe separators kept tight; raises if it outgrew the cap."""
    s = json.dumps(line, separators=(",", ":"))
    if len(s.encode()) >= LINE_CAP:
        raise ValueError(f"bench line is {len(s.encode())} bytes, cap {LINE_CAP}")
    return s
