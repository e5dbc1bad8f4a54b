"""The bench line the driver parses: the committed round-2 line (profiles/r02_bench_c2.json, printed by bench.py on
an MI355X with the driver's own command `python3 bench.py --gpus 1 --steps 20 --warmup 5`) has every key of the contract,
the metric BASELINE.json names and a self-consistent roofline object; `--gpus N` is honoured however the script is started."""
import json
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_committed_bench_line_follows_the_contract():
    line = json.loads(open(os.path.join(ROOT, "profiles", "r02_bench_c2.json")).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"] == base["metric"] and line["unit"] == "Mpixels/s"
    for key in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / the kernel's average duration; value = pixels / the same clock
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_us_avg"] * 1e-6) / 1e9) < 1.0
    assert r["algorithmic_bytes_per_launch"] == 6 * 4096 * 4096  # 3 B read + 3 B written per pixel (4:2:0)
    assert 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.10  # PMC bytes: no re-reads
    px_per_us = line["value"] * 1e6 / 1e6  # Mpixels/s -> pixels/us
    assert abs(px_per_us * line["ms_per_step"] * 1e3 - 4096 * 4096) / (4096 * 4096) < 0.01
    c = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "value_is"):
        assert key in c, key
    assert c["kind"] in ("port", "reference")
    # round 2: the K-step block is repeated, the median is the number; the traffic figure says where it comes from;
    # the other configurations are timed in the same run
    assert line["steps"] == 20 and line["warmup"] == 5 and line["blocks"] >= 15
    assert line["ms_per_step_min"] <= line["ms_per_step"] <= line["ms_per_step_max"]
    assert r["traffic_source"].startswith("profile: profiles/")
    for name in ("c3", "c2_444", "c2_unaligned", "c5"):
        assert 0.1 < line["other_configs"][name]["frac"] < 1.0, name
    assert line["whole_file"]["file_bytes"] == 11150133


def _stub_line(gpus, extra=()):
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", str(gpus), "--steps", "4", "--warmup", "1",
                        "--blocks", "3", "--no-cpu-baseline", *extra], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout  # stdout carries ONE line: the JSON (libraries' banners go to stderr)
    return json.loads(lines[0])


def test_bench_honours_gpus_when_started_without_a_launcher():
    """`python bench.py --gpus 2` (the form the driver uses for N = 1) must run TWO ranks and say so: the script
    re-executes itself under torch.distributed.run.  Plumbing only (--stub: gloo, the step is a sleep)."""
    line = _stub_line(2)
    assert line["n_gpus"] == 2 and line["data"] == "stub" and line["steps"] == 4 and line["blocks"] == 3
    assert line["ms_per_step_min"] <= line["ms_per_step"] <= line["ms_per_step_max"]
    # round 4: a run with N > 1 also measures configs[3] and configs[2] over ITS ranks and says what the process group was
    assert line["rccl"]["world"] == 2 and line["rccl"]["all_reduce_of_rank_plus_1"] == line["rccl"]["all_reduce_expected"] == 3
    assert [d["rank"] for d in line["rccl"]["devices"]] == [0, 1] and line["rccl"]["backend"] == "gloo"
    c4, c3 = line["other_configs"]["c4"], line["other_configs"]["c3_sharded"]
    for leg in (c4, c3):
        assert "error" not in leg, leg
        assert leg["n_gpus"] == 2 and leg["scaling"] == "strong" and leg["value"] > 0 and leg["ms_per_step_min"] <= leg["ms_per_step"]
    assert len(c4["config"]["file_sha256"]) == 64 and c3["config"]["images_per_rank"] == [8, 8]
    # round 5: per-rank phase breakdown of one instrumented step; the batch also in two waves
    assert len(c4["phases_ms_by_rank"]) == 2 and all("total_ms" in p for p in c4["phases_ms_by_rank"])
    assert len(c3["config"]["phases_ms_by_rank"]) == 2 and "sizes_ms" in c3["config"]["phases_ms_by_rank"][0]
    c3w = line["other_configs"]["c3_sharded_two_waves"]
    assert "error" not in c3w and c3w["config"]["waves"] == 2 and c3w["config"]["file_bytes_total"] == c3["config"]["file_bytes_total"]
    c4s = line["other_configs"]["c4_shared_arena"]  # the same file, every band's body written into one node-shared segment
    assert "error" not in c4s and c4s["config"]["file_sha256"] == c4["config"]["file_sha256"] and "shared" in c4s["config"]["workload"]
    one = _stub_line(1)
    assert one["n_gpus"] == 1 and one["rccl"]["world"] == 1 and "error" not in one["other_configs"]["c4"]
    assert one["other_configs"]["c3_sharded"]["config"]["images_per_rank"] == [16]


def test_bench_refuses_a_launcher_world_that_differs_from_gpus():
    import subprocess, sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_bench_parses_its_flags_without_a_gpu():
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], stdout=subprocess.PIPE, check=True).stdout.decode()
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--blocks", "c4", "c2_unaligned"):
        assert flag in out


def test_a_stuck_multi_gpu_leg_cannot_take_the_metric_line_with_it():
    """The legs bench.py runs after the metric's blocks (configs[3], configs[2] over the ranks) use collectives that no
    single-GPU session could exercise with N > 1: they run behind a deadline, and a leg that hangs costs its own result only —
    the ONE JSON line with the metric is still printed and the process ends with status 0."""
    import subprocess, sys
    code = r'''
import sys, time
sys.argv = ["bench.py", "--stub", "--gpus", "1", "--steps", "2", "--warmup", "1", "--blocks", "3", "--no-cpu-baseline"]
import bench
import benchlib.multi
benchlib.multi.MULTI_LEGS_DEADLINE_S = 2.0
benchlib.multi.measure_c4 = lambda *a, **k: time.sleep(120)
bench.main()
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=100)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["n_gpus"] == 1 and "abandoned" in line["other_configs"]["multi_gpu_legs"]["error"]
