OUT=gpurun_out/profiles_r04
mkdir -p $OUT
( echo "teams of 16, 3875 frames:"; python tools/ramp_profile.py 3875 16; echo "teams of 8, 3875 frames:"; python tools/ramp_profile.py 3875 8; echo "teams of 16, 10000 frames:"; python tools/ramp_profile.py 10000 16 ) 2>&1 | grep -v amdgpu.ids > $OUT/ramp_teams.txt
cat $OUT/ramp_teams.txt
