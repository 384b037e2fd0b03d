"""The multi-rank path of bench.py end to end on the one GPU of the test box: two ranks started by bench.py itself (no launcher),
both on device 0 through the rehearsal knob, gloo for the barrier and the max-over-ranks clock, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_ranks_started_by_bench_emit_one_parsed_line(gpu):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_DEVICE_OVERRIDE"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--units", "512", "--mixed-units", "384", "--steps", "2", "--warmup", "1", "--no-cpu"],
                         capture_output=True, text=True, env=env, timeout=540)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # rank 0 alone speaks on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1
    assert line["verified_bit_exact"] is True and line["value"] > 0
    assert line["config"]["units_per_gpu"] == 512
    # whole-job rate: both ranks' bytes over the slowest rank's time
    assert abs(line["value"] - 2 * 512 * 65536 / (line["ms_per_step"] * 1e-3) / 1e9) / line["value"] < 0.02
    spread = line["roofline"]["kernel_ms_avg_per_rank"]
    assert 0 < spread["min"] <= spread["max"]
    assert line["roofline"]["frac_of_n_gpus_peak"] > 0
    # configs[4] (mixed gzip + zstd, sharded) rides in every multi-GPU run at its own per-GPU size, with its own per-rank times
    assert set(line["workloads"]) == {"dynamic", "mixed"}
    mixed = line["workloads"]["mixed"]
    assert mixed["units_per_gpu"] == 384 and mixed["verified"] is True and mixed["frac_of_n_gpus_peak"] > 0
    assert 0 < mixed["kernel_ms_avg_per_rank"]["min"] <= mixed["kernel_ms_avg_per_rank"]["max"]
