cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
SELA_BENCH_FORCE_EXCHANGE=1 timeout 600 python bench.py --workload album --steps 3 --warmup 1 > gpurun_out/r02e/album1.log 2>&1
echo "album rc=$?"; tail -c 2500 gpurun_out/r02e/album1.log
SELA_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-legs > gpurun_out/r02e/track_exch.log 2>&1
echo "track+exchange rc=$?"; tail -c 600 gpurun_out/r02e/track_exch.log
