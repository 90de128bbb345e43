"""bench.py prints ONE JSON line with the driver's contract keys (run as the driver runs it: a child process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "jets/s" and d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1000 and abs(d["value"] - 256 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.3 < r["frac"] < 1.0 and (r["traffic"] is None or r["traffic"] > 1e6)
    # self-consistency: the FLOP the line says were executed per step, over the step time, cannot exceed the peak
    per_step = r["executed_flop_per_launch"] + r["executed_flop_per_train_step"]
    assert abs(per_step / (d["ms_per_step"] * 1e-3) / 1e12 - r["achieved"]) < 1e-6 * r["achieved"]
    assert per_step / (d["ms_per_step"] * 1e-3) <= r["peak"] * 1e12
    assert 0.0 < r["valid_row_fraction"] <= r["executed_row_fraction"] <= 1.0
    assert "dense_equiv_over_peak" in r and "frac_algorithmic_dense" not in r and 0.0 < r["dense_flop_not_executed_share"] < 1.0
    assert "mfma_busy" in r and (r["mfma_busy"] is None or 0.2 < r["mfma_busy"] < 1.0)
    assert "resident in HBM" in d["config"]["workload"]
    assert len(d["per_rank"]["train_ms_alone"]) == 1 and d["allreduce_ms_alone"] == 0.0


def test_two_rank_launch_rehearsal():
    """The driver's N > 1 launch line on the one-GPU box: two ranks share the card and the gradient all-reduce goes through gloo
    (PFM_BENCH_REHEARSAL; RCCL refuses two ranks on one device).  Checks the rendezvous, the per-rank control flow with the
    collective on the train stream, the max-over-ranks timing and that only rank 0 prints."""
    env = dict(os.environ, PFM_BENCH_REHEARSAL="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 512 and "REHEARSAL" in d["config"]["parallelism"]
    assert abs(d["value"] - 512 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    pr = d["per_rank"]
    assert len(pr["train_ms_alone"]) == 2 and len(pr["allreduce_ms_alone"]) == 2 and all(v > 0 for v in pr["allreduce_ms_alone"])
    assert "rendezvous" in res.stderr and "first all-reduce" in res.stderr and "first step done" in res.stderr
    # both gradient exchanges ran (23 synchronised steps each, replicas compared bitwise after each: bench.check_replicas): the
    # flat all-reduce (default) and the two-bucket exchange whose async work handles are waited for behind the queued dW GEMM
    ge = d["grad_exchange"]
    assert ge["overlapped"] is False and ge["train_ms_alone_flat_allreduce"] > 0 and ge["train_ms_alone_bucketed_overlap"] > 0
    assert "bucketed_overlap" in res.stderr and "flat_allreduce" in res.stderr
    # ... and, as the last leg of the run, the direct two-phase exchange (all-to-all of slices + local sum + all-gather; under gloo by way
    # of all_gather): timed, replicas still bit-identical, no error recorded
    tp = ge["two_phase_direct"]
    assert tp["error"] is None and tp["exchange_ms_alone"] > 0 and tp["train_ms_alone"] > 0, tp


def test_a_stage_that_hangs_ends_the_process_with_a_non_zero_code():
    """bench.py's per-rank watchdog (`stage`): a rank stuck in a stage exits 124 and says which one, instead of hanging the launcher."""
    code = ("import sys, time; sys.argv=['bench.py']; sys.path.insert(0, %r); import bench\n"
            "with bench.stage('sleeping', 0.5):\n    time.sleep(30)\n") % ROOT
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert res.returncode == 124, (res.returncode, res.stderr[-500:])
    assert "stage 'sleeping' did not finish" in res.stderr
