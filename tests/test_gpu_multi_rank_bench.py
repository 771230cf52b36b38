"""bench.py's own N-rank path with the HIP kernel behind every rank (BASELINE configs[2] shape).

`python bench.py --gpus 2 --backend gloo` is started exactly as the driver starts the bench — no
torchrun, WORLD_SIZE unset — so the parent must spawn the two ranks itself.  On the one-GPU box the
gloo rehearsal puts both ranks on cuda:0 and sends the collective through host memory; everything
else (strip partition, per-rank trace kernels, the single gather, reassembly on rank 0) is the code
the 8-GPU run executes.  The gathered 500-spp frame must equal the one-rank frame bit for bit, and a
request for more devices than the box has must fail loudly instead of rendering on one GPU."""
import json
import subprocess
import sys

import numpy as np
import pytest

import rtow
from conftest import REPO

pytestmark = pytest.mark.gpu


def run_bench(args, timeout=600):
    return subprocess.run([sys.executable, str(REPO / "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("nranks,precision", [(2, "fast"), (3, "strict")])
def test_bench_spawns_ranks_and_the_gathered_frame_is_the_one_rank_frame(ctx, tmp_path, nranks, precision):
    out = tmp_path / "frame.npy"
    r = run_bench(["--gpus", str(nranks), "--backend", "gloo", "--spp", "500", "--steps", "1", "--warmup", "1",
                   "--no-cpu-baseline", "--precision", precision, "--dump-image", str(out)])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == nranks and line["steps"] == 1
    assert "500 spp" in line["config"]["workload"] and f"over {nranks} GPUs" in line["config"]["workload"]
    assert line["config"]["spp_effective"] == 500
    assert abs(line["value"] - 1200 * 800 * 500 / (line["ms_per_step"] * 1e-3) / 1e6) / line["value"] < 1e-3
    got = np.load(out)
    scene = rtow.HostScene.cover(11, 1.5, False)
    prec = rtow.F64_FAST if precision == "fast" else rtow.F64_STRICT
    whole, st = ctx.render(scene, rtow.make_config(1200, 800, 500, 50, 50, seed=1, precision=prec))
    assert got.shape == whole.shape == (800, 1200, 3)
    assert np.array_equal(got, whole), int((got != whole).sum())


def test_bench_stdout_is_one_compact_line_and_the_details_go_to_a_file(tmp_path):
    """The driver reads the LAST line of stdout as JSON (BENCH_r04: `parsed: null` on a 20.9 KB line).  stdout of the
    bench is exactly one line under benchline.LINE_CAP with roofline and the contract's keys; everything else is in
    the details file the line names."""
    import benchline

    det = tmp_path / "details.json"
    r = run_bench(["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-scale-projection", "--details-out", str(det)])
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.endswith("\n") and r.stdout.count("\n") == 1, r.stdout[:400]
    assert len(r.stdout.encode()) < benchline.LINE_CAP
    line = json.loads(r.stdout)
    for key in benchline.CONTRACT_KEYS:
        assert key in line, key
    assert line["n_gpus"] == 1 and line["config"]["baseline_config"] == "configs[1]" and line["dtype"] == "f64"
    assert line["roofline"]["bound"] == "valu_issue" and 0 < line["roofline"]["kernel_ms"] <= line["ms_per_step"]
    assert line["details"] == str(det)
    details = json.loads(det.read_text())
    assert details["value"] == line["value"] and len(details["other_configs"]) == 4
    assert benchline.compact_line(details, str(det)) == line


def test_bench_refuses_more_gpus_than_the_box_has():
    import torch

    n = torch.cuda.device_count()
    r = run_bench(["--gpus", str(n + 7), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], timeout=120)
    assert r.returncode != 0 and "HIP device" in r.stderr and '"n_gpus"' not in r.stdout
    r = run_bench(["--gpus", "7", "--backend", "gloo", "--steps", "1", "--warmup", "0"], timeout=120)
    assert r.returncode != 0 and "at most" in r.stderr
