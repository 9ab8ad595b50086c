cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round2.py -q -k "mean_workers" -p no:cacheprovider 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -q -k "digest or 10k" -p no:cacheprovider 2>&1 | tail -2
python tools/sweep.py 1000 3875 10000 40000 2>&1 | grep -v amdgpu
python tools/phase_profile.py 2>&1 | sed -n 1,8p
