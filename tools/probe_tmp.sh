cd $GRAFT_REPO_ROOT
for k in 11 10 9 8 7 6 5 4; do echo "wgs_per_cu=$k"; SELA_DECODE_WGS_PER_CU=$k SELA_SWEEP_HOST=0 python tools/sweep.py 1000 2000 3875 10000 40000 2>&1 | grep -v "amdgpu\|frames"; done
echo auto; SELA_SWEEP_HOST=0 python tools/sweep.py 1000 2000 3875 10000 40000 2>&1 | grep -v "amdgpu\|frames"
