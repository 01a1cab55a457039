set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
export FDTD_BENCH_EXTRA=0 FDTD_BENCH_CPU=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof_pair -- python3 $R/bench.py > $O/r3_bench_1024_pair_under_rocprof.json
FDTD_PAIR=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof_single -- python3 $R/bench.py > $O/r3_bench_1024_single_step_under_rocprof.json
cd $R
timeout -k 10 400 python3 tools/pmc_traffic.py --out gpurun_out/r3_traffic_1024_pair.json --kernels k3_pair --workload "fdtd3d 1024^3 per GPU, eps/mu grids, 20-layer CPML, fp32, two steps per launch, every cell of D and B non-zero (tools/prof3d.py --mode fused --n 1024 --steps 6 --noise)" -- tools/prof3d.py --mode fused --n 1024 --steps 6 --noise > $O/r3_traffic_pair.log
FDTD_PAIR=0 timeout -k 10 400 python3 tools/pmc_traffic.py --out gpurun_out/r3_traffic_1024_fused.json --kernels k3_fused --workload "fdtd3d 1024^3 per GPU, eps/mu grids, 20-layer CPML, fp32, one step per launch, every cell of D and B non-zero (FDTD_PAIR=0 tools/prof3d.py --mode fused --n 1024 --steps 4 --noise)" -- tools/prof3d.py --mode fused --n 1024 --steps 4 --noise > $O/r3_traffic_fused.log
cd /tmp
timeout -k 10 400 python3 $R/tools/pmc_clock.py > $O/r3_pmc_clock_final.md
cd $R
timeout -k 10 300 python3 tools/bench_group.py 2 1024 > $O/r3_group_two_slabs_one_gpu.log
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof_2d -- python3 $R/bench.py --extra-leg fdtd2d_8192 > $O/r3_bench_2d_8192_under_rocprof.json
cd $R
timeout -k 10 300 python3 tools/pmc_traffic.py --out gpurun_out/r3_traffic_8192_2d_pair.json --variants --kernels k2_pair --workload "fdtd2d 8192^2, 20-layer CPML, reference Lorentz slab, fp32, two steps per launch (bench.py --extra-leg fdtd2d_8192)" -- bench.py --extra-leg fdtd2d_8192 > $O/r3_traffic_2d_pair.log
FDTD2D_PAIR=0 timeout -k 10 300 python3 tools/pmc_traffic.py --out gpurun_out/r3_traffic_8192_2d_fused.json --kernels k2_fused --workload "fdtd2d 8192^2, 20-layer CPML, reference Lorentz slab, fp32, one step per launch (FDTD2D_PAIR=0 bench.py --extra-leg fdtd2d_8192)" -- bench.py --extra-leg fdtd2d_8192 > $O/r3_traffic_2d_fused.log
timeout -k 10 300 python3 tools/bench2d_observed.py > $O/r3_bench2d_observed.json
echo measured
