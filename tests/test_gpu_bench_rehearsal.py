"""bench.py's multi-rank path rehearsed on one GPU: two ranks share cuda:0, the ghost planes travel
through gloo (host-staged) where the 8-GPU run uses RCCL.  Everything else is the code the scaling
run executes: slab engines in separate processes, the steppers' launch/exchange order, the rank-0
JSON line.  The global Dx checksum must equal the single-process run of the same grid."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(env_extra, ranks):
    env = dict(os.environ, FDTD_BENCH_N="192", FDTD_BENCH_NZ="384", FDTD_BENCH_CPU="0", **env_extra)
    if ranks == 1:
        cmd = [sys.executable, "bench.py", "--steps", "7", "--warmup", "2"]
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env.update(FDTD_BENCH_BACKEND="gloo", FDTD_BENCH_ONE_DEVICE="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", str(ranks), "--steps", "7", "--warmup", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["fused", "fused-single", "lean"])
def test_two_and_three_ranks_equal_one(mode):
    """fused: overlapping slabs, two steps per launch (what the scaling run uses); fused-single: the one-step
    kernels with their per-step plane exchanges; lean: the two-kernel path."""
    extra = {"FDTD_BENCH_MODE": mode.split("-")[0]}
    if mode == "fused-single":
        extra["FDTD_BENCH_STEPPER"] = "single"
    one = _bench(extra, 1)
    assert one["checks"]["finite"] and one["checks"]["checksum_dx"] != 0
    for ranks in (2, 3):
        many = _bench(extra, ranks)
        assert many["n_gpus"] == ranks and many["config"]["grid"] == one["config"]["grid"]
        assert many["checks"]["checksum_dx"] == one["checks"]["checksum_dx"]
        for key in ("metric", "value", "unit", "ms_per_step", "roofline", "scaling", "dtype", "config"):
            assert key in many
        _check_multi_gpu_block(many, ranks, mode, "gloo")


def _check_multi_gpu_block(line, ranks, mode, backend):
    """What makes a first run on real peers diagnosable from the one line: who took part, and per rank where
    its step went -- every kind of launch, the idle time of its stream in front of the launches that wait
    for a ghost plane, host time spent waiting for exchanges."""
    mg = line["multi_gpu"]
    assert mg["world_size"] == ranks and backend in mg["backend"] and len(mg["per_rank"]) == ranks
    roof = line["roofline"]
    assert 0 < roof["frac"] and roof["kernel_ms_per_step"] > 0 and roof["traffic"] is None
    assert roof["own_model_bytes_per_cell"] < roof["algorithmic_bytes_per_cell"] or mode == "full"
    for r, rec in enumerate(mg["per_rank"]):
        assert rec.get("rank", rec.get("slab")) == r and rec["device"] == 0
        assert rec["layers"][1] > rec["layers"][0]
        ms, n, idle = rec["launch_ms_per_step"], rec["launches_per_step"], rec["stream_idle_ms_per_step_before"]
        if mode == "fused" and "one host thread" not in backend:
            # 7 timed steps: three two-step launches (each as edge planes + bulk) and one launch of the one-step kernel
            want = {"pair_edge", "pair_interior", "fused_all"}
            assert rec["stepper"] == "PairSlabStepper"
            assert rec["stored_layers"][0] <= rec["layers"][0] and rec["stored_layers"][1] >= rec["layers"][1]
        elif mode in ("fused", "fused-single"):
            want = {"fused_edge", "fused_top", "fused_interior"} | ({"half_h_edge"} if r + 1 < ranks else set())
        else:
            want = {"half_e_interior", "half_e_edge", "half_h_interior", "half_h_edge"}
        assert want <= set(ms), (r, sorted(ms))
        # (per STEP: a two-step launch counts half)
        assert all(ms[k] > 0 and (n[k] >= 1 or (n[k] > 0 and "pair_edge" in want)) for k in want)
        assert all(v >= 0 for v in idle.values())
        assert abs(sum(ms.values()) - rec["kernel_ms_per_step"]) < 1e-6 * max(1.0, rec["kernel_ms_per_step"]) or not mode.startswith("fused")
        assert rec["first_to_last_launch_ms_per_step"] >= sum(ms.values()) * 0.999
    if "gloo" in backend:
        # the host-staged rehearsal really blocks in the exchange waits
        # (the two-step stepper exchanges once per launch: four times in seven steps)
        least = 0.5 if mode == "fused" and "one host thread" not in backend else 1
        assert all(rec["host_wait_ms_per_step"] > 0 and rec["exchange_waits_per_step"] >= least for rec in mg["per_rank"])


@pytest.mark.parametrize("mode", ["fused", "full"])
def test_single_process_group_mode_equals_one(mode):
    """`bench.py --gpus N` started as a plain process: one host thread drives N slabs through a slab group
    (here all on cuda:0).  Same grid, same checksum, same JSON keys."""
    one = _bench({"FDTD_BENCH_MODE": mode}, 1)
    env = dict(os.environ, FDTD_BENCH_N="192", FDTD_BENCH_NZ="384", FDTD_BENCH_ONE_DEVICE="1", FDTD_BENCH_MODE=mode, FDTD_BENCH_GROUP="1")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(key, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--steps", "7", "--warmup", "2"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    many = json.loads(lines[0])
    assert many["n_gpus"] == 3 and many["config"]["grid"] == one["config"]["grid"] and "one process" in many["config"]["decomposition"]
    assert many["checks"]["checksum_dx"] == one["checks"]["checksum_dx"]
    for key in ("metric", "value", "unit", "ms_per_step", "roofline", "scaling", "dtype", "config"):
        assert key in many
    _check_multi_gpu_block(many, 3, mode, "one host thread")


def test_plain_start_with_several_gpus_takes_two_steps_per_launch():
    """`python bench.py --gpus 2` without a launcher: the program starts the launcher form as a child process before
    it touches a GPU and relays the child's line -- one process per GPU, overlapping slabs, two steps per launch --
    instead of falling back to the one-step kernels of the single-process slab group."""
    one = _bench({}, 1)
    env = dict(os.environ, FDTD_BENCH_N="192", FDTD_BENCH_NZ="384", FDTD_BENCH_CPU="0", FDTD_BENCH_ONE_DEVICE="1")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FDTD_BENCH_GROUP", "FDTD_BENCH_BACKEND"):
        env.pop(key, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "7", "--warmup", "2"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]        # stdout carries the one JSON line only
    many = json.loads(lines[0])
    assert many["n_gpus"] == 2 and many["steps"] == 7 and many["warmup"] == 2
    assert many["roofline"]["kernel"] == "k3_pair" and many["roofline"]["steps_per_launch"] == 2
    assert many["checks"]["checksum_dx"] == one["checks"]["checksum_dx"]
    assert "child" in many["config"]["launched_as"]
    _check_multi_gpu_block(many, 2, "fused", "gloo")


def test_secondary_legs_ride_on_the_one_line():
    """N = 1: after the headline engine has been released the same program times the fp64 twin of the headline
    configuration, the vacuum cavity and the 2-D grid (here at test sizes) and reports them under `extra`."""
    line = _bench({"FDTD_BENCH_EXTRA": "small"}, 1)
    extra = line["extra"]
    assert set(extra) == {"fp64_small", "cavity_small", "fdtd2d_small", "fdtd2d_small_dispersive_everywhere"}
    for name, leg in extra.items():
        assert "error" not in leg, (name, leg)
        assert leg["ms_per_step"] > 0 and leg["value"] > 0 and leg["checks"]["finite"]
        assert 0 < leg["roofline"]["frac"] and leg["kernel_ms_per_step"] > 0
    assert extra["fp64_small"]["dtype"] == "f64" and extra["fp64_small"]["checks"]["sumsq_dx"] > 0
    assert extra["cavity_small"]["checks"]["sumsq_dx"] > 0
    assert extra["fdtd2d_small_dispersive_everywhere"]["dispersive_cells"] == 512 * 512
    assert 0 < extra["fdtd2d_small"]["dispersive_cells"] < 512 * 512


def test_rccl_self_exchange_on_engine_memory():
    """What one GPU can show of the RCCL path itself: the backend takes the tensors that alias the
    engine's ghost planes, and batch_isend_irecv orders correctly against the kernels on the compute
    stream (one rank exchanging with itself, see tests/helpers/nccl_self_exchange.py)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "nccl_self_exchange.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1] == "OK"
