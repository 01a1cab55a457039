"""The suite's randomized parity tests (fused, two steps per launch, the RMS window inside them, slab groups in both choreographies, two-kernel, 3-D whole grids and slab partitions, 2-D with one and with two steps per launch)
over a range of further seeds:  python tools/stress_parity.py LO HI   (needs the GPU; ~8 seeds per second)."""
import sys, time, traceback
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fused as tf, test_gpu_fused2d as t2, test_gpu_parity as tp, test_gpu_group as tg, test_gpu_pair as tq, test_gpu_observed as to


def group_two_step(seed):
    """The random slab groups with two steps per launch wherever their cuts allow it."""
    os.environ["FDTD_PAIR"] = "1"
    try:
        tg.test_group_random_configurations_equal_twin(seed, "1")
    finally:
        os.environ.pop("FDTD_PAIR", None)



def fused2d_two_step(seed):
    """The random 2-D problems with two steps per launch (rows per wave drawn from the seed), then the RMS window inside them."""
    os.environ["FDTD2D_PAIR"] = "1"
    os.environ["FDTD_Z_CHUNK2D_PAIR"] = str((1, 2, 3, 5, 8, 32)[seed % 6])
    try:
        t2.test_fused2d_random_configurations_equal_twin(seed)
        t2.test_window_random_configurations_2d(seed)
    finally:
        os.environ.pop("FDTD2D_PAIR", None)
        os.environ.pop("FDTD_Z_CHUNK2D_PAIR", None)


lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for seed in range(lo, hi):
    for name, fn in (("fused3d", tf.test_fused_random_configurations_equal_twin), ("slabs3d", tf.test_fused_random_slab_partitions_equal_twin),
                     ("group3d", lambda seed: tg.test_group_random_configurations_equal_twin(seed, None)), ("group3d two-step", group_two_step),
                     ("window3d", to.test_window_random_configurations),
                     ("pairs3d", tq.test_pairs_equal_single_steps_random_configurations)):
        try:
            fn(seed)
        except Exception as e:
            bad += 1
            print(f"FAIL {name} seed {seed}: {str(e)[:300]}", flush=True)
    for name, fn in (("fused2d", getattr(t2, "test_fused2d_random_configurations_equal_twin", None)), ("fused2d two-step", fused2d_two_step),
                     ("2d", tp.test_random_configurations_2d_equal_twin)):
        if fn is None: continue
        try:
            fn(seed)
        except Exception as e:
            bad += 1
            print(f"FAIL {name} seed {seed}: {str(e)[:300]}", flush=True)
    for mode in (0, 1):
        try:
            tp.test_random_configurations_half_step_kernels_equal_twin(seed, mode)
        except Exception as e:
            bad += 1
            print(f"FAIL half mode {mode} seed {seed}: {str(e)[:300]}", flush=True)
    if seed % 10 == 0:
        print(f"seed {seed} done, {bad} failures, {time.time()-t0:.0f}s", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
